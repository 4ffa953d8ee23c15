"""Which lane supplies / receives what in v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4 outer products)."""
import ctypes as C, os, subprocess, tempfile, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(tempfile.mkdtemp(), "m.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(root, "tools", "mfma441_map.hip"), "-o", so])
lib = C.CDLL(so)
out = torch.zeros(64 * 8, device="cuda")
assert lib.run(C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
o = out.cpu().view(64, 8)
for l in (0, 1, 2, 3, 4, 5, 31, 32, 33, 63):
    print("lane %2d: D[0..3] from A-lanes %s   from B-lanes %s" % (l, [int(x) for x in o[l, :4]], [int(x) for x in o[l, 4:]]))
