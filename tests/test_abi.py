"""CPU: the C-ABI library loads without a GPU and exports every symbol include/diffhandles_hip.h
declares; the ctypes table mirrors the header; product code has no oracle import."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "diffhandles_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from diffusionhandles_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.lib()
    assert lib.dh_missing_symbols == []
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert lib.dh_version() >= 100
    assert lib.dh_device_count() >= 0


def test_compute_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diffusionhandles_amd import depth_transform as DT
    from diffusionhandles_amd.synthetic import make_scene
    depth, bg, mask = make_scene(256)
    with pytest.raises(RuntimeError):
        DT.transform_depth(depth, bg, mask, torch.eye(3))
    from diffusionhandles_amd import DiffusionHandles
    with pytest.raises(RuntimeError):
        DiffusionHandles().to("cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "diffusionhandles_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"
                assert "/root/reference" not in src or f == "depth_transform.py", f


def test_api_surface_matches_reference_names():
    import inspect
    import diffusionhandles_amd as pkg
    from diffusionhandles_amd.depth_transform import transform_depth
    from diffusionhandles_amd.losses import compute_background_loss, compute_foreground_loss
    dh = pkg.DiffusionHandles
    for m in ("to", "invert_input_image", "generate_input_image", "set_foreground", "transform_foreground"):
        assert hasattr(dh, m)
    gd = pkg.GuidedStableDiffuser
    for m in ("to", "get_image_shape", "get_feature_shape", "init_prompt", "init_depth", "get_depth_intrinsics",
              "initial_inference", "guided_inference", "decode_latent_image", "encode_latent_image",
              "process_correspondences", "get_timesteps", "prepare_extra_step_kwargs"):
        assert hasattr(gd, m), m
    sig = inspect.signature(transform_depth)
    assert list(sig.parameters)[:4] == ["depth", "bg_depth", "fg_mask", "intrinsics"]
    assert sig.parameters["depth_transform_mode"].default == "pc"
    assert list(inspect.signature(compute_foreground_loss).parameters) == [
        "activations", "activations_orig", "processed_correspondences", "patch_size", "activations_size"]
    assert inspect.signature(compute_background_loss).parameters["loss_type"].default == "global_avg"
    sig = inspect.signature(pkg.StableNullInverter.invert)
    assert sig.parameters["num_inner_steps"].default == 10 and sig.parameters["early_stop_epsilon"].default == 1e-5
    # every public method of the reference class (stable_null_inverter.py:12-181)
    for m in ("to", "prev_step", "next_step", "get_noise_pred_single", "get_noise_pred", "latent2image", "image2latent",
              "ddim_loop", "ddim_inversion", "null_optimization", "invert"):
        assert hasattr(pkg.StableNullInverter, m), m
    sig = inspect.signature(pkg.StableNullInverter.get_noise_pred)
    assert list(sig.parameters)[1:] == ["latents", "t", "context", "depth", "is_forward"] and sig.parameters["is_forward"].default is True
    import diffhandles
    assert diffhandles.DiffusionHandles is dh


def test_ddim_scheduler_scalars_match_oracle():
    from diffusionhandles_amd.scheduler import DDIMScheduler
    from oracle import loop_ref as L
    a, b = DDIMScheduler(), L.DDIM()
    assert a.timesteps.tolist() == b.timesteps.tolist()
    for t in (0, 20, 500, 980):
        at, ap = a.step_alphas(t)
        assert at == float(b.alpha(t)) and ap == float(b.alpha(t - 20))
        af, an = a.inversion_alphas(t)
        assert af == float(b.alpha(min(t - 20, 999))) and an == float(b.alphas_cumprod[t])


def test_kernel_registers_and_scratch_audit():
    """Code-object metadata of the built library (tools/check_isa.py; no GPU): the regressions an ISA pass found in
    round 1 stay out -- kernels that silently use scratch (address-taken locals, spills), memory-bound elementwise
    kernels with so many registers that one wave fills a SIMD, workgroups that cannot launch."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    lib = os.path.join(ROOT, "diffusionhandles_amd", "libdiffhandles_hip.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(check_isa.LLVM, "llvm-readelf")):
        pytest.skip("library or llvm-readelf not present")
    ks = check_isa.kernels(lib)
    assert len(ks) > 200
    for k in ks:
        waves = (k["max_threads"] + 63) // 64
        # registers: a workgroup must fit one CU (512 unified registers per lane and SIMD, waves spread over 4 SIMDs)
        assert k["vgpr"] <= 512 // ((waves + 3) // 4), k
        assert k["lds"] <= 160 * 1024, k
        # scratch: only the known 8-byte spill of the 16-wave attention forward
        assert k["scratch"] <= (8 if "k_attn_fwd" in k["name"] else 0), k
    by = lambda frag: [k for k in ks if frag in k["name"]]
    # streaming kernels of the U-Net keep >= 4 waves per SIMD (<= 128 registers) for the row lengths SD-2 has
    for frag in ("k_ln_fwdIDF16_Li1E", "k_ln_fwdIDF16_Li2E", "k_ln_fwdIDF16_Li3E", "k_ln_bwdIDF16_Li1E", "k_ln_bwdIDF16_Li2E",
                 "k_ln_bwdIDF16_Li3E", "k_gn_apply", "k_gn_bwd_apply", "k_geglu_fwd", "k_geglu_bwd", "k_splitk_reduceI",
                 "k_gn_partial", "k_concat_gn", "k_copy_cols", "k_split_cols"):
        sel = by(frag)
        assert sel, frag
        for k in sel:
            assert k["vgpr"] <= 128, k


def test_hand_placed_vmcnt_waits_cover_their_loads():
    """tools/check_vmcnt.py (no GPU): the attention kernels issue their tile prefetch from inline asm and wait for it by hand, which
    the compiler's wait insertion does not see -- no instruction of any kernel may touch the destination register of a load that
    the in-order vmcnt scoreboard still has in flight.  First the checker itself on two hand-made instruction streams."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    import check_vmcnt
    ok = ["global_load_dwordx4 v[10:13], v[2:3], off", "global_store_dwordx4 v[4:5], v[20:23], off", "v_add_f32_e32 v1, v2, v3",
          "s_waitcnt vmcnt(1)", "v_add_f32_e32 v1, v10, v3", "s_endpgm"]
    assert check_vmcnt.check("ok", ok) == []
    early = ["global_load_dwordx4 v[10:13], v[2:3], off", "global_store_dwordx4 v[4:5], v[20:23], off", "s_waitcnt vmcnt(2)",
             "v_mov_b32_e32 v40, v12", "s_waitcnt vmcnt(0)", "s_endpgm"]
    bad = check_vmcnt.check("early", early)
    assert len(bad) == 1 and bad[0][1].startswith("v_mov_b32")
    dma = ["global_load_lds_dwordx4 v[20:21], off", "v_mov_b32_e32 v20, v3", "buffer_load_dwordx4 v7, s[4:7], s9 offen lds", "v_mov_b32_e32 v7, v3"]
    assert check_vmcnt.check("dma", dma) == []                 # LDS-DMA: the first operand is an address, nothing lands in registers
    lib = os.path.join(ROOT, "diffusionhandles_amd", "libdiffhandles_hip.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(check_isa.LLVM, "llvm-objdump")):
        pytest.skip("library or llvm-objdump not present")
    n = 0
    for name, body in check_vmcnt.kernels_disassembly(lib):
        if "k_attn_" in name or "k_gemm_pp" in name:
            n += 1
            assert check_vmcnt.check(name, body) == [], name
    assert n >= 40


def test_m0_belongs_to_the_lds_dma_statements_only():
    """The descriptor-addressed LDS-DMA of k_gemm_pp (round 5) and of the BUF instantiations of k_gemm_dma (round 6) writes M0 from
    inline asm WITHOUT saving it and without an "m0" clobber: correct only while nothing else in those kernels touches M0 (LDS
    instructions need none on gfx9+).  Audit of the disassembly (no GPU): in every such kernel M0 appears in `s_mov_b32 m0, ...`
    (the DMA statements' own writes; the address form's save / restore pairs) and nowhere else -- no s_movrel*, no v_readlane /
    v_writelane / v_interp / ds_gws / s_sendmsg operand, no read of M0 into a general register in the BUF kernels."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    import check_vmcnt
    lib = os.path.join(ROOT, "diffusionhandles_amd", "libdiffhandles_hip.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(check_isa.LLVM, "llvm-objdump")):
        pytest.skip("library or llvm-objdump not present")
    n_buf = n_pp = 0
    for name, body in check_vmcnt.kernels_disassembly(lib):
        is_pp = "k_gemm_pp" in name
        is_buf = "k_gemm_dma" in name and name.endswith("ELb1EEEvNS_5GemmKE")          # the last template argument: BUF = true
        if not (is_pp or is_buf):
            continue
        n_pp += is_pp
        n_buf += is_buf
        dma = [ln for ln in body if re.search(r"buffer_load_dwordx4 .* lds$", ln)]
        assert dma, name
        for ln in body:
            if not re.search(r"\bm0\b", ln):
                continue
            assert re.match(r"s_mov_b32 m0, (s\d+|0x[0-9a-f]+|\d+)$", ln), (name, ln)
    assert n_buf >= 60 and n_pp >= 20, (n_buf, n_pp)


def test_gemm_main_loops_carry_only_lds_dma_on_the_vector_memory_queue():
    """k_gemm_dma and k_gemm_pp wait for their LDS-DMA tiles with hand-COUNTED `s_waitcnt vmcnt(N)` (pieces per stage, plus the residual
    prefetch in front of the loop).  A count is only right while nothing else sits in the wave's in-order vector-memory queue inside the
    K loop -- a compiler-generated spill, a hoisted or sunk global access would make the wait too lax and stale LDS tiles would be
    multiplied without any fault (advisor, round 5).  Audit of the disassembly (no GPU; tools/check_loops.py): the control-flow graph of
    every kernel of the two families is rebuilt, its K loop is the innermost natural loop that contains a v_mfma (k_gemm_pp's persistent
    tile loop, prologue, residual prefetch and epilogue lie around it), and inside it the only vector-memory instructions are the LDS-DMA
    ones (`global_load_lds_dwordx4`, `buffer_load_dwordx4 ... lds`); no scratch access anywhere in those kernels."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    import check_loops
    lib = os.path.join(ROOT, "diffusionhandles_amd", "libdiffhandles_hip.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(check_isa.LLVM, "llvm-objdump")):
        pytest.skip("library or llvm-objdump not present")
    seen = {"k_gemm_dma": 0, "k_gemm_pp": 0}
    for name, body in check_loops.kernels(lib):
        fam = "k_gemm_pp" if "k_gemm_pp" in name else ("k_gemm_dma" if "k_gemm_dma" in name else None)
        if fam is None:
            continue
        assert not any(ins.startswith("scratch_") for _, ins, _ in body), name
        n_loops, n_mfma, n_dma, bad = check_loops.audit(name, body)
        assert n_loops == 1 and n_mfma >= 4 and n_dma >= 2, (name, n_loops, n_mfma, n_dma)
        assert not bad, (name, bad[:4])
        seen[fam] += 1
    assert seen["k_gemm_dma"] >= 120 and seen["k_gemm_pp"] >= 20, seen


def test_loop_audit_finds_a_stray_load_and_ignores_code_around_the_loop():
    """tools/check_loops.py on hand-made programs: a global load INSIDE the innermost MFMA loop is reported; the same load in a block
    that the outer (tile) loop runs between two K loops -- placed at a LOWER address than the K loop, reached by a jump that is not a
    back edge -- is not; an outer loop is not mistaken for the K loop."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_loops
    dma = "buffer_load_dwordx4 v1, s[0:3], s4 offen lds"
    stray = "global_load_dwordx4 v[0:3], v[4:5], off"
    bad_prog = [(0, "s_nop 0", None), (4, "v_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], a[0:15]", None), (8, dma, None), (12, stray, None),
                (16, "s_cbranch_scc1 65532", 4), (20, "s_endpgm", None)]
    n_loops, n_mfma, n_dma, bad = check_loops.audit("bad", bad_prog)
    assert (n_loops, n_mfma, n_dma) == (1, 1, 1) and bad == [("0xc", stray)]
    # entry -> jump over a cold block -> outer header -> K loop -> back to the cold block (the "prefetch for the next tile") -> outer header
    good_prog = [(0, "s_branch 2", 12),
                 (4, stray, None), (8, "s_branch 0", 12),                            # cold block at a low address, inside the OUTER loop only
                 (12, "s_nop 0", None),                                              # outer header
                 (16, "v_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], a[0:15]", None), (20, dma, None), (24, "s_cbranch_scc1 65533", 16),
                 (28, "global_store_dwordx4 v[4:5], v[0:3], off", None),             # epilogue store of the tile
                 (32, "s_cbranch_scc0 65528", 4),                                    # next tile: via the cold block
                 (36, "s_endpgm", None)]
    n_loops, n_mfma, n_dma, bad = check_loops.audit("good", good_prog)
    assert (n_loops, n_mfma, n_dma, bad) == (1, 1, 1, [])

