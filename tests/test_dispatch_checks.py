"""torch.ops.artiboost_hip.* validates its tensor arguments before the C call (ops that "validate with TORCH_CHECK"; the
round-3 review: "a wrong B overruns a buffer instead of raising").  The checks are generated from the header's `@check` contracts
(artiboost_amd/gen_torch_ops.py).  Placement / dtype violations are detected without a device; size violations need device tensors."""
import re

import pytest
import torch

from artiboost_amd import _lib as L
from artiboost_amd import gen_torch_ops as G


def ops():
    L.lib() if L.BINDING == "torch" else None
    torch.ops.load_library(L.TORCH_LIB_PATH)
    return torch.ops.artiboost_hip


def test_every_contract_names_real_arguments_and_most_pointer_ops_have_one():
    decls = {name: params for _, name, params in G.declarations()}
    cons = G.contracts()
    assert len(cons) >= 40
    for name, clauses in cons.items():
        assert name in decls
        G.checks_for(name, [p for p in decls[name] if p != ("void*", "stream")], clauses)      # asserts on unknown names / unreadable clauses
    src = open(G.OUT).read()
    for name in decls:                                       # every wrapper with a pointer argument carries placement checks
        if any(t.endswith("*") and n != "stream" for t, n in decls[name]):
            body = re.search(r"static void w_%s\(.*" % name[3:], src).group(0)
            assert "ck_dev(" in body or "ck_host(" in body or "ck_contig(" in body, name


def test_cpu_tensor_or_wrong_dtype_for_a_device_pointer_raises_before_any_launch():
    o = ops()
    f = torch.zeros(64)
    with pytest.raises(RuntimeError, match="must be a HIP tensor"):
        o.mano_lbs(f, f, f, f, f, f, f, f, 1, f, f, f)
    with pytest.raises(RuntimeError, match="must be a HIP tensor"):
        o.split_f32(torch.zeros(64), 64, torch.zeros(64, dtype=torch.bfloat16), torch.zeros(64, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match="must be a HIP tensor"):
        o.png_unfilter_batch(torch.zeros(64, dtype=torch.uint8), torch.zeros(8, dtype=torch.int32), 1, 4, 3, 4, torch.zeros(64, dtype=torch.uint8), None)
    with pytest.raises(RuntimeError, match="HOST structure"):      # a host struct handed over as a (fake) device tensor is refused as well
        o.render_batch(torch.zeros(200, dtype=torch.uint8, device="meta"), None, None, None, None, None, None, 1, 1, 8, 8, 0, None, None, None, None, None)


@pytest.mark.gpu
def test_wrong_sizes_dtypes_and_layouts_raise_runtime_error():
    o = ops()
    d = "cuda"
    bf = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device=d)      # noqa: E731
    fl = lambda *s: torch.zeros(s, dtype=torch.float32, device=d)       # noqa: E731
    N, H, W, Ci, Co = 2, 16, 16, 64, 64
    x, w, y = bf(2, N, H, W, Ci), bf(2, Co, 3, 3, Ci), fl(N, H, W, Co)
    o.conv2d_fwd_x3(x[0], x[1], w[0], w[1], y, N, H, W, Ci, Co, 3, 3, 1, 1, None, None, 0)          # the honest call passes
    with pytest.raises(RuntimeError, match="'x_hi' holds"):                                           # a batch the input does not hold
        o.conv2d_fwd_x3(x[0], x[1], w[0], w[1], fl(4, H, W, Co), 4, H, W, Ci, Co, 3, 3, 1, 1, None, None, 0)
    with pytest.raises(RuntimeError, match="'y' holds"):                                              # an output that is too small
        o.conv2d_fwd_x3(x[0], x[1], w[0], w[1], fl(1, H, W, Co), N, H, W, Ci, Co, 3, 3, 1, 1, None, None, 0)
    with pytest.raises(RuntimeError, match="must be BFloat16"):                                       # fp32 where operand planes are expected
        o.conv2d_fwd_x3(fl(N, H, W, Ci), x[1], w[0], w[1], y, N, H, W, Ci, Co, 3, 3, 1, 1, None, None, 0)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        o.conv2d_fwd_x3(bf(N, H, W, 2 * Ci)[..., ::2], x[1], w[0], w[1], y, N, H, W, Ci, Co, 3, 3, 1, 1, None, None, 0)
    with pytest.raises(RuntimeError, match="'stats' holds"):                                          # BatchNorm partial rows of another tile count
        o.conv2d_fwd_x3(x[0], x[1], w[0], w[1], y, N, H, W, Ci, Co, 3, 3, 1, 1, None, fl(1, Co, 2), 0)
    B = 3
    args = [fl(B, 48), fl(B, 10), fl(778, 3), fl(778, 3, 10), fl(778, 3, 135), fl(16, 778), fl(778, 16), fl(45)]
    o.mano_lbs(*args, B, fl(B, 778, 3), fl(B, 21, 3), fl(B, 16, 4, 4))
    with pytest.raises(RuntimeError, match="'verts' holds"):                                          # an output sized for another batch
        o.mano_lbs(fl(B + 1, 48), fl(B + 1, 10), *args[2:], B + 1, fl(B, 778, 3), fl(B + 1, 21, 3), None)
    with pytest.raises(RuntimeError, match="'pose' holds"):
        o.mano_lbs(*args, 64, fl(64, 778, 3), fl(64, 21, 3), None)
    with pytest.raises(RuntimeError, match="must be Float"):
        o.mano_lbs(args[0].double(), *args[1:], B, fl(B, 778, 3), fl(B, 21, 3), None)
    with pytest.raises(RuntimeError, match="'uvd' holds"):                                            # soft-argmax: C of another head
        o.softargmax3d_fwd(fl(2, 8, 8, 22 * 32), 0, 2, 22, 28, 32, 8, 8, fl(2, int(o.softargmax3d_ntiles(8, 8)), 22, 8), fl(2, 21, 3), fl(2, 22), fl(2, 22, 2))
    with pytest.raises(RuntimeError, match="must be BFloat16"):                                       # dtype code and tensor disagree
        o.softargmax3d_fwd(fl(2, 8, 8, 22 * 32), 1, 2, 22, 28, 32, 8, 8, fl(2, int(o.softargmax3d_ntiles(8, 8)), 22, 8), fl(2, 22, 3), fl(2, 22), fl(2, 22, 2))
    with pytest.raises(RuntimeError, match="'m' holds"):                                              # optimizer state shorter than the parameter
        o.clip_adam_x3(fl(1024), fl(1024), fl(512), fl(1024), 1024, fl(1), 0.001, 5e-5, 0.9, 0.999, 1e-8, 1, None, bf(1024), bf(1024))
    with pytest.raises(RuntimeError, match="bytes"):                                                  # a workspace of the wrong launch
        o.augment_batch(torch.zeros(2, 32, 32, 4, dtype=torch.uint8, device=d), 2, 32, 32, torch.zeros(2, 4, dtype=torch.int32, device=d), fl(2, 4),
                        fl(2, 6), None, None, 16, 16, 0, fl(2, 22, 24, 4), None, torch.zeros(16, dtype=torch.uint8, device=d))
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_functional_forms_allocate_their_outputs_and_match_the_pointer_forms():
    from artiboost_amd import kernels as K
    o = ops()
    g = torch.Generator().manual_seed(2)
    x = K.split(torch.randn(2, 16, 16, 64, generator=g).cuda())
    w = K.split((torch.randn(64, 3, 3, 64, generator=g) * 0.05).cuda())
    y_ref, st_ref = K.conv2d_fwd_x3(x, w, 1, 1, want_stats=True)
    y, st = o.conv2d_fwd_x3.fn(x, w, 1, 1, None, True, False)
    assert torch.equal(y, y_ref) and torch.equal(st, st_ref)
    y2, st2 = o.conv2d_fwd_x3.fn(x, w, 2, 1)
    assert y2.shape == (2, 8, 8, 64) and st2.numel() == 0 and torch.equal(y2, K.conv2d_fwd_x3(x, w, 2, 1))
    with pytest.raises(RuntimeError, match="planes"):
        o.conv2d_fwd_x3.fn(x[0], w, 1, 1)
    B = 5
    pose, betas = (torch.randn(B, 48, generator=g) * 0.3).cuda(), (torch.randn(B, 10, generator=g)).cuda()
    tabs = [torch.randn(s, generator=g).cuda() * 0.01 for s in ((778, 3), (778, 3, 10), (778, 3, 135))] + [torch.rand(16, 778, generator=g).cuda(),
            torch.softmax(torch.randn(778, 16, generator=g), 1).cuda(), torch.zeros(45).cuda()]
    v, j, T = o.mano_lbs.fn(pose, betas, *tabs)
    v2, j2, T2 = torch.empty_like(v), torch.empty_like(j), torch.empty_like(T)
    o.mano_lbs(pose, betas, *tabs, B, v2, j2, T2)
    assert v.shape == (B, 778, 3) and j.shape == (B, 21, 3) and T.shape == (B, 16, 4, 4)
    assert torch.equal(v, v2) and torch.equal(j, j2) and torch.equal(T, T2)
    with pytest.raises(RuntimeError, match=r"betas must be \[B, 10\]"):
        o.mano_lbs.fn(pose, betas[:2], *tabs)
