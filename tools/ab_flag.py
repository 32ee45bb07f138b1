"""A/B of a module-level flag on ONE box: run bench.py with `module.FLAG = value` set before main().
    python tools/ab_flag.py onda_amd.ops FUSE_BN_FINALIZE 0 -- --steps 20 --warmup 5 --no-eager ...   (value: a Python literal)
The flag must EXIST in the module (a typo would otherwise measure nothing)."""
import ast, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mod, flag, val = sys.argv[1:4]
m = importlib.import_module(mod)
assert hasattr(m, flag), f"{mod} has no attribute {flag!r}: nothing to A/B"
setattr(m, flag, ast.literal_eval(val))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[sys.argv.index("--") + 1:]
import bench  # noqa: E402
bench.main()
