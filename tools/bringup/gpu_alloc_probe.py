"""bring-up: alignment of hipMalloc / torch allocations (do large scratch buffers start on 2 MiB boundaries?)"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
torch.cuda.init(); x = torch.empty(16 << 30, dtype=torch.uint8, device="cuda"); y = torch.empty((16 << 30) + 1078542379 - (16 << 30) % 7, dtype=torch.uint8, device="cuda")
print("torch 16 GiB: %x (mod 2M = %x)   torch odd size: %x (mod 2M = %x)" % (x.data_ptr(), x.data_ptr() & 0x1FFFFF, y.data_ptr(), y.data_ptr() & 0x1FFFFF))
for sz in (1811939328 + 64, (8 << 30) + 64, 100003, (384 << 10) * 4608 + 64, 3 << 20):
    p = ctypes.c_void_p()
    r = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(sz))
    print("hipMalloc %12d -> rc %d addr %x  mod 2M = %x  mod 64K = %x" % (sz, r, p.value or 0, (p.value or 0) & 0x1FFFFF, (p.value or 0) & 0xFFFF))
