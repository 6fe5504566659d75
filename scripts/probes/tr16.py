import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtr16.so"))
def run(addrs, name):
    a = torch.tensor(addrs, dtype=torch.int32, device="cuda")
    o = torch.zeros(64 * 4, dtype=torch.int16, device="cuda")
    rc = lib.run_probe(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(o.data_ptr()))
    r = o.cpu().view(64, 4).tolist()
    print("==", name, "rc", rc)
    for l in range(64):
        print(l, addrs[l] // 2, r[l])
# pattern A: linear, lane*8 bytes
run([l * 8 for l in range(64)], "linear lane*8B (word index = lane*4)")
# pattern B: 4x16 block of a row-major [row][64 words] matrix: lane i of a 16-lane group -> row i//4, col (i%4)*4 ; groups -> +16 cols
run([((l % 16) // 4) * 128 + ((l % 16) % 4) * 8 + (l // 16) * 32 for l in range(64)], "block 4 rows x 16 cols, row stride 128B, group g at col 16g")
# pattern C: every lane its own distinct far-apart address (lane*128 bytes = row lane, col 0)
run([l * 128 for l in range(64)], "lane -> row lane, col 0")
