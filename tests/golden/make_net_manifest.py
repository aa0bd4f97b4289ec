"""Record the reference's MyBigFRFNModel (= --arch NewBigFRFNModel, src/model/model.py:465-583) surface: state_dict keys, shapes and
per-tensor sums of its default initialisation under torch.manual_seed(1234).  Build container only (stand-ins: _ref_stubs.py).

    python tests/golden/make_net_manifest.py   ->  tests/golden/frfnnet_manifest.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402

_ref_stubs.install()
from src.model.model import MyBigFRFNModel  # noqa: E402

torch.manual_seed(1234)
m = MyBigFRFNModel()
sd = m.state_dict()
np.savez_compressed(os.path.join(HERE, "frfnnet_manifest.npz"),
                    names=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
                    sums=np.array([float(v.double().sum()) for v in sd.values()]),
                    n_params=np.int64(sum(p.numel() for p in m.parameters())))
print(len(sd), "tensors,", sum(p.numel() for p in m.parameters()), "parameters")
