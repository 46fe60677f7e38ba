#!/usr/bin/env python3
"""Facts about page-locked host memory on this platform (for the soak's one GPU fault at a host-heap address, profiles/r05_soak.txt):
is the device pointer of a hipHostRegister'ed numpy array its host address?  does torch see it as pinned?  what about torch's own
pinned blocks?  is a copy from pageable memory asynchronous to the host?"""
import ctypes as C, time
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
torch.cuda.init(); torch.zeros(1, device="cuda")
def devptr(hostptr):
    out = C.c_void_p()
    rc = hip.hipHostGetDevicePointer(C.byref(out), C.c_void_p(hostptr), 0)
    hip.hipGetLastError()                                  # (a failed query must not stay behind as the runtime's last error)
    return rc, out.value
for n in (768, 12 * 64 * 64, 12 * 512 * 512, 12 * 1024 * 1024):
    a = np.random.rand(n)
    p = a.ctypes.data
    rc = torch.cuda.cudart().cudaHostRegister(p, a.nbytes, 0)
    r2, d = devptr(p)
    print(f"numpy {a.nbytes:>10d} B at {p:#x}: register rc {int(rc)}, device pointer rc {r2} {d if d is None else hex(d)} same={d == p} "
          f"torch.is_pinned {torch.from_numpy(a).is_pinned()}")
    t = torch.empty(n, dtype=torch.float64, device="cuda")
    t.copy_(torch.from_numpy(a), non_blocking=True); torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), a)
    torch.cuda.cudart().cudaHostUnregister(p)
    r3, d3 = devptr(p)
    print(f"   after unregister: device pointer rc {r3}, torch.is_pinned {torch.from_numpy(a).is_pinned()}")
h = torch.empty(1 << 20, dtype=torch.float64, pin_memory=True)
r, d = devptr(h.data_ptr())
print(f"torch pinned block at {h.data_ptr():#x}: device pointer rc {r} {hex(d)} same={d == h.data_ptr()}")
# pageable source: does the call return before the data has been read?
a = np.ones(12 * 1024 * 1024); t = torch.empty(a.size, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for nb in (True, False):
    t0 = time.perf_counter(); t.copy_(torch.from_numpy(a), non_blocking=nb); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"pageable 101 MB copy_(non_blocking={nb}): call {1e3*(t1-t0):.2f} ms, then sync {1e3*(t2-t1):.2f} ms")
a[:] = 2.0                      # overwritten right after an asynchronous call returned: what arrived?
t.copy_(torch.from_numpy(a), non_blocking=True); a[:] = 3.0; torch.cuda.synchronize()
print("pageable non_blocking copy, source overwritten after the call returned: device holds", set(np.unique(t.cpu().numpy()).tolist()))
