"""The code object hipRTC builds for a workload's user lambdas, disassembled (no GPU needed: sdpgpu_create_custom compiles before
it asks for a device):  python tools/custom_functor_isa.py custom_clsp [out.s]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "custom_clsp"
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/%s.s" % name
os.environ["SDPGPU_CUSTOM_DUMP"] = out + ".co"
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
w = workloads.by_name(name)
try:
    sia.SdpEngine(w.desc(), w.pmf, w.overhead(), custom_source=w.custom_source, custom_params=w.custom_params)
except Exception as e:  # (no device here: the handle is not created, the code object is already written)
    print("engine:", str(e)[:120])
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
with open(out, "w") as f:
    subprocess.run([objdump, "-d", out + ".co"], stdout=f, check=True)
print(out, sum(1 for _ in open(out)), "lines")
