"""Stand-alone timing of the stable radix sort on (cell | depth)-like 40-bit keys (HIP events, many repetitions)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, "/root/repo/ad-gs_amd")
from adgs import _lib
lib = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_944_258
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.RandomState(0)
depth = (rng.rand(n).astype(np.float32) * 78 + 2).view(np.uint32).astype(np.uint64)
keys = (rng.randint(0, 150, size=n).astype(np.uint64) << np.uint64(32)) | depth
kd = torch.from_numpy(keys.view(np.int64)).cuda(); vd = torch.arange(n, dtype=torch.int32, device="cuda")
ko = torch.empty_like(kd); vo = torch.empty_like(vd)
tmp = torch.empty(int(lib.adgs_test_sort_temp_bytes(n)), dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(): assert lib.adgs_test_sort_pairs_u64(kd.data_ptr(), ko.data_ptr(), vd.data_ptr(), vo.data_ptr(), n, bits, tmp.data_ptr(), st) == 0
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 50
order = np.argsort(keys & np.uint64((1 << bits) - 1), kind="stable")
ok = np.array_equal(ko.cpu().numpy().view(np.uint64), keys[order]) and np.array_equal(vo.cpu().numpy().view(np.uint32), order.astype(np.uint32))
print("n=%d bits=%d: %.1f us per sort (%.1f us per pass), %.2f Gpairs/s, correct=%s" % (n, bits, ms * 1e3, ms * 1e3 / ((bits + 7) // 8), n / ms / 1e6, ok))
