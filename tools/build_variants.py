"""Build library variants for A/B runs into build/ab/: python tools/build_variants.py name:-DFLAG,-DFLAG2 ...   (LCQP_VARIANT_NCH=8 in the
environment builds the kernels of another padded size than np = 256)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
from concurrent.futures import ThreadPoolExecutor
vars_ = {}
for a in sys.argv[1:]:
    name, _, defs = a.partition(":")
    vars_[name] = [d for d in defs.split(",") if d]
def one(kv):
    name, defs = kv
    os.makedirs(os.path.join(ROOT, "build/ab"), exist_ok=True)
    out = os.path.join(ROOT, "build/ab", name + ".so")
    g.build_hip(force=True, out=out, defines=defs, only_nch=int(os.environ.get("LCQP_VARIANT_NCH", "2")))
    return out
with ThreadPoolExecutor(2) as ex:
    for o in ex.map(one, vars_.items()):
        print("built", o)
