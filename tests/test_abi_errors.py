"""Error behaviour of the C ABI on a CPU-only box: every entry point validates its arguments on the host before anything touches
a GPU, returns a negative code and leaves a message in the thread-local dlwp_last_error().  Round-2 entries."""
import ctypes as C
import os

import pytest


@pytest.fixture(scope="module")
def h():
    from dlwp_benchmark_amd import lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return L.load()


FAKE = 0x1000      # a non-NULL pointer value: validation must fail on the SHAPES before it is ever dereferenced


def err(h):
    return h.dlwp_last_error().decode()


def test_gemm_mixed_rejects_bad_dtypes_and_bf16_accumulation(h):
    assert h.dlwp_gemm_mixed(FAKE, FAKE, FAKE, 8, 8, 8, 8, 8, 8, 0, 1, None, 0, None, None, 0, None, 16, None) < 0
    assert "mask" in err(h)
    assert h.dlwp_gemm_mixed(FAKE, FAKE, FAKE, 8, 8, 8, 8, 8, 8, 0, 1, None, 0, None, None, 1, None, 4, None) < 0
    assert "fp32 output" in err(h)
    assert h.dlwp_gemm_mixed(None, FAKE, FAKE, 8, 8, 8, 8, 8, 8, 0, 1, None, 0, None, None, 0, None, 0, None) < 0
    assert h.dlwp_gemm_mixed(FAKE, FAKE, FAKE, 0, 8, 8, 8, 8, 8, 0, 1, None, 0, None, None, 0, None, 0, None) < 0


def test_window_advance_rejects_inconsistent_shapes(h):
    # batch stride smaller than ctx * frame
    assert h.dlwp_window_advance_fwd(FAKE, 10, FAKE, FAKE, FAKE, 2, 3, 16, 0, 0, 0, 0, 0, 0, None) < 0
    assert "batch stride" in err(h)
    # patch layout whose D*H*W does not match the frame
    assert h.dlwp_window_advance_fwd(FAKE, 48, FAKE, FAKE, FAKE, 2, 3, 16, 1, 1, 4, 5, 2, 1, None) < 0
    assert "patch layout" in err(h)
    assert h.dlwp_window_advance_fwd(None, 48, FAKE, FAKE, FAKE, 2, 3, 16, 0, 0, 0, 0, 0, 0, None) < 0
    assert h.dlwp_window_advance_bwd(None, None, 0, None, 0, None, None, 2, 3, 16, 0, 0, 0, 0, 0, 0, None) < 0
    assert "g_delta" in err(h)
    assert h.dlwp_window_advance_bwd(None, None, 0, FAKE, 16, None, FAKE, 0, 3, 16, 0, 0, 0, 0, 0, 0, None) < 0


def test_data_movement_entries_reject_null_and_empty(h):
    assert h.dlwp_patch_merge(None, FAKE, 1, 4, 4, 8, 0, None) < 0
    assert h.dlwp_patch_merge(FAKE, FAKE, 1, 0, 4, 8, 0, None) < 0
    assert h.dlwp_upconv_shuffle(FAKE, None, None, FAKE, None, 1, 4, 4, 8, 2, 2, 4, 0, 1, 0, None) < 0      # ctot < coff + O
    assert h.dlwp_upconv_shuffle(FAKE, None, None, FAKE, None, 1, 4, 4, 8, 2, 2, 8, 0, 1, 1, None) < 0      # backward without gout
    assert h.dlwp_add_bcast(FAKE, None, FAKE, 2, 16, None) < 0
    assert h.dlwp_cast_bf16(None, FAKE, 16, None) < 0
    assert h.dlwp_cast_bf16(0x1004, FAKE, 16, None) < 0                                                     # misaligned source
    assert "aligned" in err(h)
    assert h.dlwp_window_attn_pack_table(FAKE, None, 10, 2, 2, None) < 0
    assert h.dlwp_layernorm_fwd_ex(None, FAKE, FAKE, FAKE, FAKE, FAKE, 4, 8, C.c_float(1e-5), 1, None) < 0
    assert h.dlwp_layernorm_bwd_ex(FAKE, FAKE, FAKE, FAKE, None, 1, None, FAKE, FAKE, FAKE, 4, 8, None) < 0


def test_fft_plan_rejects_bad_shapes(h):
    out = C.c_void_p()
    assert h.dlwp_fft_plan_create(0, 8, C.byref(out)) < 0
    assert h.dlwp_fft_plan_create(8, 1, C.byref(out)) < 0
    assert h.dlwp_fft_plan_create(8, 8, None) < 0


def test_window_attention_rejects_bad_shapes(h):
    # nW must divide B_, ntypes must divide nW, head_dim <= 64 in the fused kernels
    args = lambda B_, nW, N, TB, nt, heads, d: (FAKE, FAKE, None, FAKE, FAKE, None, FAKE, FAKE, B_, nW, N, TB, nt, heads, d,  # noqa: E731
                                               C.c_float(0.1), None)
    assert h.dlwp_window_attn_fwd_packed(*args(5, 2, 49, 169, 1, 4, 16)) < 0
    assert h.dlwp_window_attn_fwd_packed(*args(4, 2, 49, 169, 3, 4, 16)) < 0
    assert "ntypes" in err(h)
    assert h.dlwp_window_attn_fwd_packed(*args(4, 2, 49, 169, 1, 4, 96)) < 0
    assert "head_dim" in err(h)


def test_round3_entries_reject_bad_arguments(h):
    """Round-3 entries: planar / block-planar transforms, the block-planar AFNO weight image, pad-aware window gather."""
    # no plan / NULL buffers (a plan cannot be built without a GPU: its tables live in device memory)
    assert h.dlwp_rfft2_planar(None, FAKE, FAKE, FAKE, 1, 8, 0, 4, 3, 0, 1, 0, None) < 0
    assert "rfft2_planar" in err(h)
    assert h.dlwp_irfft2_planar(None, FAKE, FAKE, FAKE, None, 1, 8, 0, 4, 3, 0, 1, 0, None) < 0
    assert "irfft2_planar" in err(h)
    assert h.dlwp_afno_wq_expand_bp(None, FAKE, FAKE, FAKE, 4, 8, 8, None) < 0
    assert h.dlwp_afno_wq_expand_bp(FAKE, FAKE, FAKE, FAKE, 4, 1, 8, None) < 0          # the bias rides on the first 2 nb bs_out elements
    assert "block size" in err(h)
    assert h.dlwp_afno_wq_fold_bp(FAKE, None, FAKE, FAKE, 4, 8, 8, None) < 0
    assert h.dlwp_afno_wq_fold_bp(FAKE, FAKE, FAKE, FAKE, 0, 8, 8, None) < 0
    assert h.dlwp_weight_grad_group(None, 2, None) < 0
    assert h.dlwp_weight_grad_group(FAKE, 0, None) < 0
    assert h.dlwp_set_gemm_tile256(2) < 0


def test_token_layout_attention_entries_validate_before_launching(h):
    """round 4: dlwp_window_attn_fwd_tokens / _bwd_tokens -- NULL operands, inconsistent window counts, query ranges, and (on a box
    without the bf16 matrix mode set) the unsupported-shape path return negative codes with a message."""
    F = C.c_float
    # forward: qkv, fill, table, packed, ia, ib, labels, src_map, dst_map, out, lse, B_, nW, N, Ltok, TB, ntypes, heads, d, scale, q_lo, q_hi, io, stream
    assert h.dlwp_window_attn_fwd_tokens(None, FAKE, FAKE, None, FAKE, FAKE, None, FAKE, FAKE, FAKE, FAKE, 8, 4, 49, 100, 169, 1, 2, 16, F(0.25), 0, 49, 0, None) < 0
    assert "NULL" in err(h)
    assert h.dlwp_window_attn_fwd_tokens(FAKE, FAKE, FAKE, None, FAKE, FAKE, None, FAKE, FAKE, FAKE, FAKE, 9, 4, 49, 100, 169, 1, 2, 16, F(0.25), 0, 49, 0, None) < 0
    assert "bad shape" in err(h)                                               # B_ is not a multiple of nW
    assert h.dlwp_window_attn_fwd_tokens(FAKE, FAKE, FAKE, None, FAKE, FAKE, None, FAKE, FAKE, FAKE, FAKE, 8, 4, 49, 100, 169, 1, 2, 16, F(0.25), 10, 60, 0, None) < 0
    assert "query range" in err(h)
    assert h.dlwp_window_attn_fwd_tokens(FAKE, FAKE, FAKE, None, FAKE, FAKE, None, FAKE, FAKE, FAKE, FAKE, 8, 4, 49, 100, 169, 1, 2, 16, F(0.25), 0, 49, 0, None) < 0
    assert "2048" in err(h) or "bf16 matrix mode" in err(h)                    # 16 (window, head) pairs: not this family's shape
    # backward: qkv, fill, table, packed, ia, ib, labels, out, lse, gout, dst_map, src_map, gqkv, gfill, gtable, B_, nW, N, Ltok, TB, ...
    assert h.dlwp_window_attn_bwd_tokens(FAKE, None, FAKE, None, FAKE, FAKE, None, FAKE, FAKE, FAKE, None, FAKE, FAKE, FAKE, FAKE, 8, 4, 98, 100, 2548, 1, 2, 32,
                                         F(0.2), 0, 98, 0, None) < 0
    assert "NULL" in err(h)
    assert h.dlwp_window_attn_bwd_tokens(FAKE, None, FAKE, None, FAKE, FAKE, None, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, 8, 4, 98, 100, 2548, 1, 2, 32,
                                         F(0.2), 0, 98, 1, None) < 0
    assert "fill" in err(h)                                                    # bf16 tensors without the token-layout operands
    assert h.dlwp_window_attn_bwd_tokens_supported(200, 32, 2548) == 0 and h.dlwp_window_attn_bwd_tokens_supported(98, 48, 2548) == 0
    assert h.dlwp_window_attn_fwd_tokens_supported(98, 32, 100) == 0           # too few (window, head) pairs
