"""Record the reference CLI's argparse defaults (args.py:6-279) as a fixture: the namespace `main.py:17-18,69-70` hands to
``ModelTrainer.train(args, args.arch, args.max_epoch, args.lr)`` when the user passes no flags.  Run in the build
container only (the reference never travels); args.py imports nothing but argparse.

    python tests/golden/make_args_defaults.py   ->  tests/golden/ref_args_defaults.json
"""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_args", "/root/reference/args.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
ns = vars(mod.argument_parser().parse_args([]))
with open(os.path.join(HERE, "ref_args_defaults.json"), "w") as f:
    json.dump(ns, f, indent=1, sort_keys=True)
print(len(ns), "defaults recorded")
