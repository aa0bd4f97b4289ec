"""Per-kernel counter sums of tools/pmc_fused.sh: for every (kernel, grid size) the mean over its dispatches.  python tools/pmc_fused_sum.py <outdir>"""
import collections
import csv
import glob
import re
import sys

out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(uir_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:60]
        key = (name, int(r["Grid_Size"]))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
for key in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    c = {k: acc[key][k] / cnt[key][k] for k in acc[key]}
    n = cnt[key].get("SQ_WAVES", 1)
    w = c.get("SQ_WAVES", 1)
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print(f"{key[0][:60]:60s} grid {key[1]:8d} x{n:3d}  waves {w:7.0f}  cycles/wave {wc / w:9.0f}  VALU/wave {c.get('SQ_INSTS_VALU', 0) / w:7.0f}  MFMA/wave {c.get('SQ_INSTS_MFMA', 0) / w:6.0f}  "
          f"LDS/wave {c.get('SQ_INSTS_LDS', 0) / w:6.0f}  wait_any {c.get('SQ_WAIT_INST_ANY', 0) / wc:5.2f}  wait_lds {c.get('SQ_WAIT_INST_LDS', 0) / wc:5.2f}  "
          f"lds_busy/gui {c.get('SQ_LDS_IDX_ACTIVE', 0) / max(1.0, c.get('GRBM_GUI_ACTIVE', 1)):7.2f}  bank_conf/lds_busy {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 1)):5.2f}  "
          f"mfma_busy/gui {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1.0, c.get('GRBM_GUI_ACTIVE', 1)):7.2f}  vmem_rd/wave {c.get('SQ_INSTS_VMEM_RD', 0) / w:6.0f}  vmem_wr/wave {c.get('SQ_INSTS_VMEM_WR', 0) / w:6.0f}")
