"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
include/pmoe_hip.h declares, with the argument counts the ctypes binding uses (no compute calls)."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

from pmoe_amd import hip

REPO = Path(__file__).resolve().parents[1]
HEADER = (REPO / "include" / "pmoe_hip.h").read_text()


def _declared():
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    text = re.sub(r"typedef struct.*?\}\s*\w+;", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:int|int64_t|const char\*)\s+(pmoe_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


@pytest.fixture(scope="module")
def lib():
    if not hip.lib_path().exists():
        import __graft_entry__
        __graft_entry__.build()
    return hip.load()


def test_header_declares_the_bound_symbols():
    decl = _declared()
    assert set(decl) == set(hip.SIGNATURES), set(decl) ^ set(hip.SIGNATURES)
    for name, n in decl.items():
        assert len(hip.SIGNATURES[name]) == n, (name, n, len(hip.SIGNATURES[name]))


def test_library_exports_every_symbol(lib):
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.pmoe_version() == hip.ABI_VERSION == 401
    assert b"invalid" in lib.pmoe_error_string(-1)


def test_struct_layouts_match_c(lib):
    # sizes computed by hand from include/pmoe_hip.h (LP64): 6 pointers + 22 int32 + float + pad + u64 + 2 int32 + float +
    # pad + pointer + int32 + pad + pointer + int32 + pad
    assert ctypes.sizeof(hip.ConvDesc) == 200 == lib.pmoe_abi_sizeof(0)
    assert ctypes.sizeof(hip.WgradDesc) == lib.pmoe_abi_sizeof(1)
    # int32 block padded to 8, part_ws + its size, grads + 3 int32 + bn_fused, 4 bn_* pointers, bn_z_ld (+ pad)   (round 4: 144 -> 184)
    assert ctypes.sizeof(hip.WgradDesc) == 3 * 8 + 19 * 4 + 4 + 2 * 8 + 8 + 4 * 4 + 4 * 8 + 4 + 4 == 184
    from pmoe_amd import optim
    assert ctypes.sizeof(optim.OptTensor) == 64 == lib.pmoe_abi_sizeof(2)
    assert optim.CHUNK == 16384          # PMOE_OPT_CHUNK


def test_no_cpu_fallback(lib):
    import torch
    from pmoe_amd import ops
    x = torch.zeros(1, 4, 4, 16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.maxpool_fwd(x, torch.zeros(1, 2, 2, 16), torch.zeros(1, 2, 2, 16, dtype=torch.uint8))


def test_conv_plan_reports_the_kernel_instantiation():
    """pmoe_conv2d_plan (bench.py uses it to attribute launch times to rocprof kernel symbols)."""
    import ctypes as C
    from pmoe_amd.hip import ConvDesc, load

    def plan(cin, cout, H, ks, stride, dtype, B=64, E=4, dilate=False, Hout=None):
        d = ConvDesc()
        pad = ks // 2
        Ho = Hout or (H + 2 * pad - ks) // stride + 1
        d.n, d.h, d.w_, d.cin, d.ho, d.wo, d.cout, d.coutp = E * B, H, H, cin, Ho, Ho, cout, (cout + 63) // 64 * 64
        d.in_ld, d.out_ld, d.ipe, d.ks, d.stride, d.pad, d.dilate = cin, cout, B, ks, stride, pad, int(dilate)
        d.dtype = 0 if dtype == torch.bfloat16 else 1
        return load().pmoe_conv2d_plan(C.byref(d))
    assert plan(64, 64, 128, 3, 1, torch.bfloat16) == 1207                 # conv3x3_respipe_kernel<false, 0> (filter bank resident, LDS-DMA patches)
    assert plan(16, 64, 256, 3, 1, torch.bfloat16) == 1316                 # 12(16)-channel stem: conv3x3_c16_kernel (direct form)
    assert plan(256, 256, 32, 3, 1, torch.bfloat16) == 5057                # conv3x3_dma_stream_kernel<true> (LDS-DMA staged, 16x16x32 MFMA shape from 256 input channels, persistent + producer wave)
    assert plan(128, 128, 64, 3, 1, torch.bfloat16) == 5047                # conv3x3_dma_stream_kernel<false> (32x32x16)
    assert plan(128, 256, 64, 1, 2, torch.bfloat16) == 1404                # 1x1 stride 2 (downsample): conv1x1_direct_kernel<4>
    assert plan(256, 512, 32, 1, 2, torch.bfloat16) == 1402                # ... 64-channel slabs from 256 input channels
    assert plan(256, 256, 32, 3, 1, torch.float32) == 722                  # f32: 4-wave 128x128 tile, 32-channel chunks
    assert plan(1536, 512, 1, 1, 1, torch.bfloat16) == 3000                # expert MLP GEMM: gemm_skinny_kernel
    assert plan(1536, 512, 1, 1, 1, torch.float32) == 722                  # ... in f32: generic 4-wave tile
    assert plan(512, 512, 14, 3, 1, torch.bfloat16, B=1, E=3) == 3000      # B=1 inference layer4: tap-looping skinny kernel
    assert plan(128, 128, 56, 3, 1, torch.bfloat16, B=1, E=3) == 3000      # ... layer2 (3136 pixels per expert <= 3200)
    assert plan(128, 128, 64, 3, 1, torch.bfloat16, B=1, E=3) == 5047      # 4096 pixels per expert: the LDS-DMA tile
    assert plan(128, 64, 64, 3, 1, torch.bfloat16, dilate=True, Hout=128) == 4741   # stride-2 dgrad, 64 gradient rows: 4 class launches <7,4,1>
    assert plan(256, 128, 32, 3, 1, torch.bfloat16, dilate=True, Hout=64) == 9207   # ... >= 128 rows: 4 class launches of conv3x3s2_dma_kernel<true>
    assert plan(64, 128, 128, 3, 2, torch.bfloat16) == 5207                # stride-2 forward: conv3x3s2_dma_kernel<false>
    assert plan(128, 64, 64, 3, 1, torch.float32, dilate=True, Hout=128) // 1000 == 4   # ... in f32: 4 class launches of the generic kernel


def test_planning_calls_see_the_same_descriptor_as_the_launch():
    """The kernel choice can depend on the row lengths (32-bit offset ranges): a planning descriptor that leaves them 0 means
    DENSE rows, and must agree with the one the launch gets -- at E=8, B=64 (BASELINE config 3's shard) the 16-channel stem
    convolution once sized its BatchNorm partial-sum buffer for a different kernel than the one that ran."""
    import ctypes as C
    from pmoe_amd.hip import ConvDesc, load

    def desc(E, B, cin, cout, H, in_ld, out_ld):
        d = ConvDesc()
        d.n, d.h, d.w_, d.cin, d.ho, d.wo, d.cout, d.coutp = E * B, H, H, cin, H, H, cout, 64
        d.in_ld, d.out_ld, d.ipe, d.ks, d.stride, d.pad, d.dtype, d.in_shared = in_ld, out_ld, B, 3, 1, 1, 0, 1
        return d
    for E, B in ((4, 64), (8, 64), (1, 1)):
        a, b = desc(E, B, 16, 64, 256, 0, 0), desc(E, B, 16, 64, 256, 16, 64)
        assert load().pmoe_conv2d_plan(C.byref(a)) == load().pmoe_conv2d_plan(C.byref(b)) == 1316
        assert load().pmoe_conv2d_stat_rows(C.byref(a)) == load().pmoe_conv2d_stat_rows(C.byref(b)) > 0
    # rows 16x as long (a window of a wide buffer): one expert's images no longer fit 32-bit offsets -> another kernel, and the
    # planning call says so
    wide = desc(8, 64, 16, 64, 256, 16, 1024)
    assert load().pmoe_conv2d_plan(C.byref(wide)) == 1005


def test_wgrad_plan_reports_the_kernel_instantiation():
    """pmoe_conv2d_wgrad_plan: the LDS-DMA weight-gradient kernel for the dense 3x3 bf16 layers, its 2 x 4 wave layout for layers
    with <= 32 input channels (the stem's first convolution), the register-staged kernel for the strided ones."""
    import ctypes as C
    from pmoe_amd.hip import WgradDesc, load

    def plan(cin, cinp, cout, H, ks, stride, B=64, E=4, per_image=0):
        d = WgradDesc()
        pad = ks // 2
        Ho = (H + 2 * pad - ks) // stride + 1
        d.n, d.h, d.w_, d.cin, d.cinp, d.ho, d.wo, d.cout, d.coutp = E * B, H, H, cin, cinp, Ho, Ho, cout, (cout + 63) // 64 * 64
        d.x_ld, d.dy_ld, d.ipe, d.ks, d.stride, d.pad, d.dtype, d.per_image = cin, cout, B, ks, stride, pad, 0, per_image
        return load().pmoe_conv2d_wgrad_plan(C.byref(d))
    assert plan(64, 64, 64, 128, 3, 1) == 7309          # conv_wgrad_dma2_kernel (round 4: one accumulating wave per SIMD + a request-only wave)
    assert plan(256, 256, 256, 32, 3, 1) == 7309
    assert plan(16, 64, 64, 256, 3, 1) == 7109
    assert 6000 <= plan(64, 64, 128, 128, 3, 2) < 7000


def test_launch_recorder_is_transparent_for_planning_calls():
    """hip.LaunchRecorder (pmoe_amd/infer.py:PlannedMixture): planning / query entry points pass through unrecorded, and the
    proxy is only in place while a recorder is active."""
    import ctypes as C
    from pmoe_amd import hip
    if torch.cuda.is_available():
        pytest.skip("the recorder's stream query is exercised by tests/test_model_gpu.py on a GPU")
    lib = hip.load()
    assert not isinstance(lib, hip._RecordingLib)
    rec = hip.LaunchRecorder()
    hip._recorder = rec                        # (entering the context needs a GPU stream)
    try:
        proxy = hip.load()
        assert isinstance(proxy, hip._RecordingLib)
        assert proxy.pmoe_abi_sizeof(0) == C.sizeof(hip.ConvDesc)
        assert rec.calls == []
    finally:
        hip._recorder = None
    assert hip.load() is lib


def test_cycle_stamped_tools_build_compiles(tmp_path):
    """tools/stamp_conv.py's -DPMOE_STAMP variant of the dominant conv kernel (never the product build) keeps compiling."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    for name in ("conv_igemm", "conv_dma", "conv_wgrad"):
        src = REPO / "pmoe_amd" / "csrc" / f"{name}.hip"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DPMOE_STAMP",
                               "-c", str(src), "-o", str(tmp_path / f"{name}_stamp.o")], stderr=subprocess.DEVNULL)
    assert (tmp_path / "conv_igemm_stamp.o").stat().st_size > 0


def test_buffer_stores_with_sgpr_offset_keep_their_data_registers(tmp_path):
    """gfx950 reads the data registers of a > 64-bit buffer store after the instruction has issued; hipcc pads the ISA's
    "store data -> VALU write" hazard only for stores WITHOUT an SGPR offset (round 3: with one, conv3x3_respipe_kernel stored
    lanes 12-15 / 44-47 of a dword that the next instruction had already overwritten).  Static guard over the kernels that use
    raw buffer stores: no VALU write to a store's data registers within two instructions of a store that carries an SGPR offset."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    for name in ("conv_c16", "conv_c1x1", "conv_res"):
        out = tmp_path / f"{name}.s"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               str(REPO / "pmoe_amd" / "csrc" / f"{name}.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
        lines = out.read_text().split("\n")
        for n, l in enumerate(lines):
            m = re.match(r"\s*buffer_store_dwordx[34] v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\], (s\d+)", l)
            if not m:
                continue
            lo, hi = int(m.group(1)), int(m.group(2))
            k, seen = n + 1, 0
            while seen < 2 and k < len(lines):
                t = lines[k].strip()
                k += 1
                if not t or t[0] in ";.":
                    continue
                seen += 1
                if t.startswith("s_nop"):
                    break
                w = re.match(r"v_(?!cmp)\S+ v\[?(\d+)", t)
                assert not (w and lo <= int(w.group(1)) <= hi), (name, l.strip(), t)


def test_vmcnt_counting_kernels_have_no_scratch_in_their_main_loops(tmp_path):
    """the LDS-DMA kernels order their operands with COUNTED `s_waitcnt vmcnt(N)`; a register spill would put scratch
    loads / stores -- which count in vmcnt too -- into the stream and let an MFMA read a tile that has not landed, silently.
    Static guard on the ISA: conv3x3_dma_kernel<*> (incl. the producer-wave instantiations), conv3x3s2_dma_kernel<*>,
    conv3x3_dma_f8_kernel and conv3x3_respipe_kernel<*> have NO scratch access at all; conv_wgrad_dma2_kernel's request-only
    waves may spill set-up values once (they wait for vmcnt(0) anyway), but nothing between its accumulating waves' MFMAs."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.split("\n")[0]

    def kernels(name):
        out = tmp_path / f"{name}.s"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               str(REPO / "pmoe_amd" / "csrc" / f"{name}.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
        text = out.read_text()
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
            yield m.group(1), m.group(2).split("\n")
    checked = 0
    for src, pats in (("conv_dma", ("conv3x3_dma_kernel", "conv3x3_dma_stream_kernel", "conv3x3s2_dma_kernel", "conv3x3_dma_f8_kernel")),
                      ("conv_res", ("conv3x3_respipe_kernel",))):
        for sym, body in kernels(src):
            if any(p in sym for p in pats):
                bad = [l.strip() for l in body if "scratch_" in l]
                assert not bad, (ver, sym, bad[:4])
                checked += 1
    for sym, body in kernels("conv_wgrad"):
        if "conv_wgrad_dma2_kernel" in sym:
            mf = [k for k, l in enumerate(body) if "v_mfma" in l]
            assert mf and not [l for l in body[mf[0]:mf[-1] + 1] if "scratch_" in l], (ver, sym)
            checked += 1
    assert checked >= 10, checked
