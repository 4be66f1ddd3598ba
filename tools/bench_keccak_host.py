import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
L = _lib.lib()
zk.sumcheck._decl()
n = 256 << 20
buf = np.frombuffer(np.random.default_rng(1).bytes(n), np.uint8).copy()
out = np.zeros(32, np.uint8)
for _ in range(2):
    t0 = time.perf_counter()
    L.zk_keccak256(_lib.p8(buf), n, _lib.p8(out))
    dt = time.perf_counter() - t0
    print(f"keccak256 host: {n/dt/1e9:.3f} GB/s", flush=True)
import subprocess
print(subprocess.run("lscpu | grep -E 'Model name|MHz|Flags' | cut -c1-300", shell=True, capture_output=True, text=True).stdout)
