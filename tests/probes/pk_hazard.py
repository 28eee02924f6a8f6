"""Packed-f32 cross-swizzle probe (tests/hip/pk_hazard.hip): the standalone kernel at one and two workgroups per CU,
with and without a matrix burst between tiles.  Prints, per configuration, how many sums of each variant differ from the scalar sums:
    own-dst   forced v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0], destination in registers of its own
    plain-C   whatever the compiler formed from `xa + xb` (cross-swizzled packed adds: tools/pk_scan.py tests/hip/libvadx_testhooks.so)
    dst=swz   forced, destination = the swizzled source's register pair
    dst=pln   forced, destination = the plain source's register pair
    war       forced, the sources overwritten by the instructions that follow at once
usage (GPU box): python tests/probes/pk_hazard.py"""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch  # noqa: E402

h = C.CDLL(os.path.join(R, "tests", "hip", "libvadx_testhooks.so"))
h.vadx_test_pk_hazard.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]


def run(nblocks, tiles, burst, lds, reps=3):
    n = nblocks * tiles * 512 + 576
    g = torch.Generator(device="cuda").manual_seed(7)
    audio = (torch.randn((16, n), device="cuda", generator=g) * 0.1).contiguous()
    mism = torch.zeros(8, dtype=torch.int32, device="cuda")
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(reps):
        assert h.vadx_test_pk_hazard(audio.data_ptr(), audio.stride(0), nblocks, tiles, burst, lds, mism.data_ptr(), sink.data_ptr(), st) == 0
    torch.cuda.synchronize()
    return [int(x) for x in mism.cpu()], reps * nblocks * tiles * 512 * 32


h.vadx_test_lds_order.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]


def run_order(nblocks, tiles, burst, lds, mode=0, reps=3):
    n = nblocks * tiles * 512 + 576
    g = torch.Generator(device="cuda").manual_seed(9)
    audio = (torch.randn((16, n), device="cuda", generator=g) * 0.1).contiguous()
    mism = torch.zeros(8, dtype=torch.int32, device="cuda")
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(reps):
        assert h.vadx_test_lds_order(audio.data_ptr(), audio.stride(0), nblocks, tiles, burst, lds, mode, mism.data_ptr(), sink.data_ptr(), st) == 0
    torch.cuda.synchronize()
    return [int(x) for x in mism.cpu()], reps * nblocks * tiles * 512


if __name__ == "__main__":
    for lds in (80 * 1024, 160 * 1024):
        for burst in (0, 64, 512):
            for nblocks, tiles in ((512, 8), (8192, 2)):
                m, tot = run_order(nblocks, tiles, burst, lds, 1)
                print(f"FRESH lds={lds:6d} ({'2 wg/CU' if lds <= 81920 else '1 wg/CU'}) burst={burst:4d} grid={nblocks:5d} x {tiles}: threads x tiles={tot:10d} "
                      f"packed sums != scalar: behind lgkmcnt(4)={m[0]} behind lgkmcnt(2)={m[1]} after lgkmcnt(0)={m[2]}", flush=True)
                m, tot = run_order(nblocks, tiles, burst, lds)
                print(f"ORDER lds={lds:6d} ({'2 wg/CU' if lds <= 81920 else '1 wg/CU'}) burst={burst:4d} grid={nblocks:5d} x {tiles}: threads x tiles={tot:10d} "
                      f"early behind lgkmcnt(4)={m[0]} lgkmcnt(2)={m[1]} lgkmcnt(1)={m[2]} sentinel={m[3]}", flush=True)
    for lds in (80 * 1024, 160 * 1024, 65536):
        for burst in (0, 64, 512):
            for nblocks, tiles in ((256, 8), (512, 8), (2048, 4), (8192, 2)):
                m, tot = run(nblocks, tiles, burst, lds)
                print(f"lds={lds:6d} ({'2 wg/CU' if lds <= 81920 else '1 wg/CU'}) burst={burst:4d} grid={nblocks:5d} x {tiles} tiles: sums={tot:11d} "
                      f"differing own-dst={m[0]} plain-C={m[1]} dst=swz={m[2]} dst=pln={m[3]} war={m[4]}", flush=True)
