"""How does the CPU oracle scale over the host's cores?  (bench.py's cpu_baseline: 128 pinned workers reach 12 x one worker.)
Runs orc_synth_bench with different worker placements: packed into one L3 domain, spread one per L3 domain, all cores.
usage: python tools/micro/cpu_scaling.py [per_worker=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O

per = int(sys.argv[1]) if len(sys.argv) > 1 else 4
phys, allc = O.host_cpu_topology()
l3 = {}
for c in phys:
    try:
        key = open(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list").read().strip()
    except OSError:
        key = "all"
    l3.setdefault(key, []).append(c)
doms = list(l3.values())
print(f"{len(phys)} physical cores, {len(allc)} hardware threads, {len(doms)} L3 domains of {len(doms[0])} cores")
o = O.default_options(perturbStep=0, printLevel=0)
def run(tag, cpus):
    ok, sec, _, _, _ = O.synth_bench(0, len(cpus), per, cpus=cpus, opt=o, want_xy=False)
    print(f"{tag:44s} workers {len(cpus):4d}  {len(cpus) * per / sec:8.1f} LCQPs/s  = {len(cpus) * per / sec / len(cpus):6.2f} per worker  ({sec:.2f} s, {ok} solved)", flush=True)
run("one worker", phys[:1])
run("2 workers in one L3 domain", doms[0][:2])
run("4 workers in one L3 domain", doms[0][:4])
run("all cores of one L3 domain", doms[0])
run("one worker per L3 domain", [d[0] for d in doms])
run("two workers per L3 domain", [c for d in doms for c in d[:2]])
run("four workers per L3 domain", [c for d in doms for c in d[:4]])
run("all physical cores", phys)
