import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbuflds.so"))
src = torch.arange(1024, dtype=torch.int32, device="cuda") + 1000
out = torch.zeros(512, dtype=torch.int32, device="cuda")
nbytes = 64 * 16  # only the first 64 units (4 KB would be 256) are inside the resource: 1024 B
rc = lib.run_probe(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), nbytes)
o = out.cpu().view(-1, 4)[:64]
print("rc", rc)
for l in (0, 1, 31, 32, 33, 34, 35, 61, 62, 63):
    print(l, [hex(v & 0xffffffff) if v < 0 or v > 100000 else v for v in o[l].tolist()])
