"""The compiler's register / LDS / occupancy report of the dense kernels at np = 256 (hipcc -Rpass-analysis=kernel-resource-usage with the
flags of __graft_entry__.HIP_FLAGS), for the standard build and for the 256-register build of the persistent kernels (-DLCQP_TU_FEW).
    python tools/kernel_resources.py > profiles/<round>/final/kernel_resource_usage.txt        (no GPU needed)"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

KEEP = ("VGPRs:", "AGPRs:", "ScratchSize", "Occupancy", "SGPRs Spill", "VGPRs Spill", "LDS Size")


def report(extra):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + g.HIP_FLAGS + ["--cuda-device-only", "-c", "-o", os.path.join(tmp, "x.co"),
               os.path.join(g.CSRC, "lcqp_nch.hip"), "-DLCQP_TU_NCH=2", "-Rpass-analysis=kernel-resource-usage"] + extra
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    filt = shutil.which("c++filt")
    out, cur = [], None
    for l in err.splitlines():
        m = re.search(r"remark: +(.*?) \[-Rpass-analysis", l)
        if not m: continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            nm = t.split(": ", 1)[1]
            if filt: nm = subprocess.run([filt, nm], capture_output=True, text=True).stdout.strip() or nm
            cur = [nm]; out.append(cur)
        elif cur is not None and t.startswith(KEEP):
            cur.append(t)
    return out


print("Compiler resource usage of the np = 256 translation unit (hipcc -Rpass-analysis=kernel-resource-usage, flags of __graft_entry__.HIP_FLAGS; tools/kernel_resources.py).")
print("Waves per SIMD = 512 / (VGPRs + AGPRs), rounded down; a 256-thread workgroup is one wave per SIMD, so this is also the workgroups per CU the registers allow.")
print("(The VGPR column of rocprofv3 kernel traces in this directory shows about half of these figures.)")
print("\n-- standard build (lcqp_nch.hip -DLCQP_TU_NCH=2)")
for o in report([]): print(o[0] + "\n    " + "; ".join(o[1:]))
print("\n-- second build of the persistent kernels for batches of at most three workgroups per CU (-DLCQP_TU_FEW=1 -DLCQP_VARIANT=1 -DLCQP_MINWAVES=2)")
for o in report(["-DLCQP_TU_FEW=1", "-DLCQP_VARIANT=1", "-DLCQP_MINWAVES=2"]): print(o[0] + "\n    " + "; ".join(o[1:]))
