"""-m gpu: HIP path vs the CPU oracle and the reference-generated golden fixtures, through the C ABI."""
import os

import numpy as np
import pytest
import torch

from parity_util import assert_close, attribute_detection_diffs, greedy_violations, nms_layout, rel_err, report

pytestmark = pytest.mark.gpu

VOX_CASES = ["tiny", "edges", "cloud_cap", "coarse", "all_out"]


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------------------------------------ voxelizer
@pytest.mark.parametrize("case", VOX_CASES)
def test_voxelizer_matches_reference_golden(hip, golden, case):
    """Bit-exact (ints) / exact (copied floats) against the reference's points_to_voxel outputs."""
    from futuredet_amd.voxelize import points_to_voxel

    g = golden("voxelizer.npz")
    cfg = g[case + "_cfg"]
    vs, rg, mp, mv = cfg[:3], cfg[3:9], int(cfg[9]), int(cfg[10])
    v, c, n = points_to_voxel(g[case + "_points"], vs, rg, mp, True, mv)
    assert np.array_equal(c, g[case + "_coors"])
    assert np.array_equal(n, g[case + "_num"])
    assert np.array_equal(v, g[case + "_voxels"])


def test_voxelizer_fused_mean_and_batch_column(hip, golden):
    g = golden("voxelizer.npz")
    cfg = g["coarse_cfg"]
    out = hip.voxelize(_dev(g["coarse_points"]), cfg[:3], cfg[3:9], int(cfg[9]), int(cfg[10]), batch_idx=3,
                       want_voxels=False, want_mean=True, mean_stride=16, coor_cols=4)
    m = int(out["num_voxels"].cpu()[0])
    ref_v, ref_n = g["coarse_voxels"], g["coarse_num"]
    assert m == len(ref_n)
    mean = out["mean"][:m].cpu().numpy()
    ref_mean = ref_v.sum(1) / ref_n[:, None].astype(np.float32)  # voxel_encoder.py:17-24
    np.testing.assert_allclose(mean[:, :5], ref_mean, rtol=1e-6, atol=1e-6)
    assert np.all(mean[:, 5:] == 0)
    co = out["coors"][:m].cpu().numpy()
    assert np.all(co[:, 0] == 3) and np.array_equal(co[:, 1:], g["coarse_coors"])


def test_voxelizer_full_size_vs_oracle(hip):
    """300k-point synthetic cloud, config grid, 160k cap hit: identical to the sequential oracle."""
    from futuredet_amd.synth import synthetic_cloud
    from futuredet_amd.voxelize import points_to_voxel
    from oracle import ops as oops

    pts = synthetic_cloud(seed=0, target_points=300000)
    vs, rg = [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0]
    for cap in (160000, 120000):  # eval cap, train cap (preprocess.py:256-258); the second one is hit
        v, c, n = points_to_voxel(pts, vs, rg, 10, True, cap)
        ov, oc, on = oops.points_to_voxel(pts, vs, rg, 10, True, cap)
        assert np.array_equal(c, oc) and np.array_equal(n, on) and np.array_equal(v, ov)
    assert len(on) == 120000, "the synthetic 300k cloud is expected to hit the 120k voxel cap"


# ------------------------------------------------------------------------------------------------ index / rulebook
def _random_sparse(rng, B, D, H, W, p, cin):
    occ = rng.random((B, D, H, W)) < p
    idx = np.argwhere(occ).astype(np.int32)
    rng.shuffle(idx)
    feats = rng.standard_normal((len(idx), cin)).astype(np.float32)
    return idx, feats


GEOMS = [((3, 3, 3), (1, 1, 1), (1, 1, 1), True), ((3, 3, 3), (2, 2, 2), (1, 1, 1), False),
         ((3, 3, 3), (2, 2, 2), (0, 1, 1), False), ((3, 1, 1), (2, 1, 1), (0, 0, 0), False)]


@pytest.mark.parametrize("geom", GEOMS)
def test_rulebook_matches_oracle_pairs(hip, geom):
    """Same (input coord, output coord, tap) triples as the spconv-1.0 restatement (row order is free)."""
    from oracle import ops as oops

    ks, st, pd, subm = geom
    rng = np.random.default_rng(7)
    B, D, H, W = 2, 11, 21, 19
    idx, _ = _random_sparse(rng, B, D, H, W, 0.12, 4)
    o_idx, pairs, pnum, oshape = oops.rulebook(idx, (D, H, W), ks, st, pd, subm)
    src = hip.SparseIndex(B, D, H, W, torch.device("cuda"))
    n_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    src.mark(_dev(idx))
    src.scan(n_dev)
    src.finalize(int(n_dev.cpu()[0]))
    assert src.n == len(idx)
    if subm:
        dst = src
        kk, ss, pp = ks, (1, 1, 1), tuple(k // 2 for k in ks)
    else:
        dst = src.downsample(ks, st, pd)
        nd = torch.zeros(1, dtype=torch.int32, device="cuda")
        dst.scan(nd)
        dst.finalize(int(nd.cpu()[0]))
        kk, ss, pp = ks, st, pd
    assert dst.spatial_shape == list(oshape)
    assert dst.n == len(o_idx)
    co_in, co_out = src.coords.cpu().numpy(), dst.coords.cpu().numpy()
    assert set(map(tuple, co_out)) == set(map(tuple, o_idx))
    # rows are sorted by the tiled column order: strictly increasing column key, z ascending inside a column
    nbr = src.rulebook(dst, kk, ss, pp).cpu().numpy()
    assert np.all(nbr[:, dst.n:] == -1)
    got = set()
    for k in range(nbr.shape[0]):
        for o in np.nonzero(nbr[k, :dst.n] >= 0)[0]:
            got.add((tuple(co_in[nbr[k, o]]), tuple(co_out[o]), k))
    want = set()
    for k in range(len(pnum)):
        for t in range(pnum[k]):
            want.add((tuple(idx[pairs[k, 0, t]]), tuple(o_idx[pairs[k, 1, t]]), k))
    assert got == want


@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_spconv_apply_vs_oracle(hip, cin, cout, dtype):
    """fd_spconv_apply (+bias +residual +relu) vs the oracle's pair-list indice_conv, element-wise
    |d| <= tol * max(1, |ref|): fp32 1e-4 (north_star allows 1e-3; an fp32 FMA chain of 27*cin terms lands near 1e-6),
    bf16 (config 3) 2e-2.  Every kernel variant is run: fp32 = the pair-compacting kernel (default) and the
    register-resident kernel at 1 / 2 / 4 row groups per wave; bf16 = the LDS-shared-weights kernel (default; 1-4 row groups per
    wave x ring depth 2 / 4), the column-split kernel and the register kernel.  Variants of one kernel agree bit for bit;
    different kernels agree within the tolerance."""
    from oracle import ops as oops

    rng = np.random.default_rng(cin * 7 + cout)
    B, D, H, W = 2, 9, 40, 37
    idx, feats = _random_sparse(rng, B, D, H, W, 0.2, cin)
    w = (rng.standard_normal((27, cin, cout)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    src = hip.SparseIndex(B, D, H, W, torch.device("cuda"))
    n_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    src.mark(_dev(idx))
    src.scan(n_dev)
    src.finalize(int(n_dev.cpu()[0]))
    row_of = src.lookup(_dev(idx))
    x = hip.rows_permute(_dev(feats), row_of, cin, tdt, n_rows=src.n)
    res_np = rng.standard_normal((src.n, cout)).astype(np.float32)
    res = _dev(res_np).to(tdt)
    nbr = src.rulebook(src, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    wpk = hip.pack_spconv_weight(torch.from_numpy(w), tdt).cuda()
    run = lambda: hip.spconv_apply(x, wpk, _dev(bias), nbr, src.n, cout, residual=res, relu=True).float().cpu().numpy()  # noqa: E731
    v1_knob = "spconv_v1" if dtype == "f32" else "spconv_bf16_v1"
    y_ws, y_ring, y_ring_v = [], None, []
    try:
        y_default = run()
        if dtype == "bf16":
            # the default kernel family of the shape -- the LDS-window kernel (fd_spconv_bf16win.hip) for 64 -> 64 and 128 -> 128,
            # the round-3 RESIDENT / RING kernels (fd_spconv_bf16.hip) otherwise: rows per wave x gather-ring depth variants must agree
            # bit for bit with the default (fixed summation order: taps ascending, channel chunks ascending)
            for rg in (1, 2, 3, 4):
                for depth in (2, 4):
                    hip.set_tuning("bf16_rg", rg)
                    hip.set_tuning("bf16_depth", depth)
                    y_ws.append(run())
            hip.set_tuning("bf16_rg", 0)
            hip.set_tuning("bf16_depth", 0)
            if cin == cout and cin >= 64:  # the RING kernels behind the window kernel: their own variants, bit for bit
                hip.set_tuning("bf16_win", -1)
                y_ring = run()
                for rg in (1, 2, 3):
                    hip.set_tuning("bf16_rg", rg)
                    y_ring_v.append(run())
                hip.set_tuning("bf16_rg", 0)
            hip.set_tuning("bf16_gp", -1)  # the older kernels: column split (default where it applies) ...
            y_old = run()
        hip.set_tuning(v1_knob, 1)          # ... and the register kernel
        ys = []
        for rg in (1, 2, 4):  # every row-group variant of the register kernel
            hip.set_tuning("spconv_rg", rg)
            ys.append(run())
    finally:
        hip.set_tuning("spconv_rg", 0)
        hip.set_tuning(v1_knob, 0)
        hip.set_tuning("bf16_rg", 0)
        hip.set_tuning("bf16_depth", 0)
        hip.set_tuning("bf16_gp", 0)
        hip.set_tuning("bf16_win", 0)
    for other in y_ring_v:
        assert np.array_equal(y_ring, other), "bf16 RING / RESIDENT kernels: rows-per-wave variants must agree bit for bit"
    for other in y_ws:
        assert np.array_equal(y_default, other), "bf16: rows-per-wave / ring-depth variants must agree bit for bit"
    for other in ys[1:]:
        assert np.array_equal(ys[0], other), "row-group variants must agree bit for bit (same fma chain per row)"
    if dtype == "f32" and cin == 16:
        # 16 input channels default to the resident-weights kernel (fd_spconv_f32r.hip; one and two row groups per wave agree bit
        # for bit); the pair-compacting kernel is the other fp32 family for this shape
        try:
            hip.set_tuning("f32_res_rg", 2 if cout == 16 else 32)  # (16 -> 32 defaults to the compacting kernel: 32 forces the resident one)
            y_rg2 = run()
            hip.set_tuning("f32_res_rg", -1)
            y_compact = run()
        finally:
            hip.set_tuning("f32_res_rg", 0)
        if cout == 16:
            assert np.array_equal(y_default, y_rg2), "resident fp32 kernel: row-group variants must agree bit for bit"
        else:
            assert_close("spconv_apply f32 16->32 resident kernel vs compacting kernel", y_rg2, y_compact, 1e-4)
    if dtype == "f32" and (cin, cout) == (32, 32):
        # 32 -> 32 has two pair-compacting kernels: 32-pair items on the 32x32x2 MFMA (default) and 16-pair items on 16x16x4
        try:
            hip.set_tuning("spconv_c32", -1)
            ys.append(run())
        finally:
            hip.set_tuning("spconv_c32", 0)
    # oracle in original row order, mapped to index order
    if dtype == "bf16":
        feats = torch.from_numpy(feats).bfloat16().float().numpy()
        w = torch.from_numpy(w).bfloat16().float().numpy()
        res_np = torch.from_numpy(res_np).bfloat16().float().numpy()
    o_idx, pairs, pnum, _ = oops.rulebook(idx, (D, H, W), (3, 3, 3), (1, 1, 1), (1, 1, 1), True)
    ref = oops.indice_conv(feats, w, bias, pairs, pnum, len(o_idx))
    r = row_of.cpu().numpy()
    ref_sorted = np.empty_like(ref)
    ref_sorted[r] = ref
    ref_sorted = np.maximum(ref_sorted + res_np, 0)
    tol = 1e-4 if dtype == "f32" else 2e-2
    assert_close("spconv_apply %s %d->%d default kernel vs oracle" % (dtype, cin, cout), y_default, ref_sorted, tol)
    assert_close("spconv_apply %s %d->%d register kernel vs oracle" % (dtype, cin, cout), ys[0], ref_sorted, tol)
    if dtype == "bf16":
        assert_close("spconv_apply bf16 %d->%d round-2 kernel vs oracle" % (cin, cout), y_old, ref_sorted, tol)
        if y_ring is not None:
            assert_close("spconv_apply bf16 %d->%d RING / RESIDENT kernel (behind the window kernel) vs oracle" % (cin, cout), y_ring, ref_sorted, tol)
    if dtype == "f32" and (cin, cout) == (32, 32):
        assert_close("spconv_apply f32 32->32 16-pair compacting kernel vs oracle", ys[-1], ref_sorted, tol)
    if dtype == "f32" and cin == 16:
        assert_close("spconv_apply f32 %d->%d pair-compacting kernel vs oracle" % (cin, cout), y_compact, ref_sorted, tol)


@pytest.mark.parametrize("c", [32, 64, 128])
def test_bf16_window_kernel_without_locality_multi_pass_and_device_count(hip, c):
    """fd_spconv_bf16win.hip (64, 128 channels; 32 runs the same checks on the RESIDENT kernel of fd_spconv_bf16.hip): correctness must not depend on the rulebook's locality (an item with a neighbour outside the LDS window
    takes the global gather), on the number of passes of a workgroup, or on where the row count comes from.  (a) a SubM rulebook of a
    sorted index, (b) the same rulebook with its input rows PERMUTED at random (no locality at all: every item gathers), (c) the same with
    the row count read from device memory and a capacity-sized launch -- all against a float64 host evaluation of
    sum_k in[nbr[k][o]] @ W[k] on the bf16-rounded operands (2e-2), (b) and (c) bit-identical to (a); row-group variants of (b) bit for
    bit (58k rows: at one row group per wave the 256 workgroups need two passes each)."""
    rng = np.random.default_rng(c)
    B, D, H, W = 1, 21, 96, 96  # ~58k rows: more than 256 workgroups x 128 rows, so one row group per wave needs two passes
    idx, feats = _random_sparse(rng, B, D, H, W, 0.3, c)
    src = hip.SparseIndex(B, D, H, W, torch.device("cuda"))
    n_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    src.mark(_dev(idx))
    src.scan(n_dev)
    src.finalize(int(n_dev.cpu()[0]))
    n = src.n
    x = hip.rows_permute(_dev(feats), src.lookup(_dev(idx)), c, torch.bfloat16, n_rows=n)
    w = (rng.standard_normal((27, c, c)) * np.sqrt(2.0 / (27 * c))).astype(np.float32)
    bias = rng.standard_normal(c).astype(np.float32)
    wpk = hip.pack_spconv_weight(torch.from_numpy(w), torch.bfloat16).cuda()
    nbr = src.rulebook(src, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    res = _dev(rng.standard_normal((n, c)).astype(np.float32)).bfloat16()
    y_a = hip.spconv_apply(x, wpk, _dev(bias), nbr, n, c, residual=res, relu=True)
    # float64 evaluation of the definition on the rounded operands (torch matmul in double precision: not one of our kernels)
    xf, wf = x.double(), torch.from_numpy(w).bfloat16().double().cuda()
    ref = _dev(bias).double().repeat(n, 1)
    for k in range(27):
        m = nbr[k, :n] >= 0
        ref[m] += xf[nbr[k, :n][m].long()] @ wf[k]
    ref = torch.clamp_min(ref + res.double(), 0).cpu().numpy()
    assert_close("bf16 window kernel %d->%d vs float64 evaluation of the definition (%d rows)" % (c, c, n), y_a.float().cpu().numpy(), ref, 2e-2)
    # (b) permuted input rows: x2[perm[i]] = x[i], rulebook entries renamed -- the same sums, no locality
    perm = torch.from_numpy(rng.permutation(n)).cuda()
    x2 = torch.empty_like(x)
    x2[perm] = x
    nbr2 = nbr.clone()
    sel = nbr2[:, :n] >= 0
    nbr2[:, :n][sel] = perm[nbr2[:, :n][sel].long()].int()
    for a in ("n_out", "n_dev", "n_expected"):
        if hasattr(nbr, a):
            setattr(nbr2, a, getattr(nbr, a))
    y_b = hip.spconv_apply(x2, wpk, _dev(bias), nbr2, n, c, residual=res, relu=True)
    assert torch.equal(y_a, y_b), "a rulebook without locality must give the same bits (every item on the gather path)"
    try:
        for rg in (1, 2, 4):
            hip.set_tuning("bf16_rg", rg)
            assert torch.equal(y_b, hip.spconv_apply(x2, wpk, _dev(bias), nbr2, n, c, residual=res, relu=True))
            assert torch.equal(y_a, hip.spconv_apply(x, wpk, _dev(bias), nbr, n, c, residual=res, relu=True))
    finally:
        hip.set_tuning("bf16_rg", 0)
    # (c) capacity launch with the row count on the device (what a captured sweep does): rows beyond the count are not written
    cap = n + 777
    nbr3 = torch.full((27, (cap + 63) // 64 * 64), -1, dtype=torch.int32, device="cuda")
    nbr3[:, :n] = nbr[:, :n]
    nbr3.n_out, nbr3.n_dev, nbr3.n_expected = cap, torch.tensor([n], dtype=torch.int32, device="cuda"), 2000  # (a low estimate: several passes)
    x3 = torch.cat([x, torch.zeros((cap - n, c), dtype=x.dtype, device="cuda")])
    res3 = torch.cat([res, torch.zeros((cap - n, c), dtype=x.dtype, device="cuda")])
    out3 = torch.full((cap, c), 7.0, dtype=torch.bfloat16, device="cuda")
    hip.spconv_apply(x3, wpk, _dev(bias), nbr3, cap, c, residual=res3, relu=True, out=out3)
    assert torch.equal(out3[:n], y_a) and bool((out3[n:] == 7.0).all())
    report("bf16 window kernel %d->%d: permuted rulebook, row-group variants and a device-count capacity launch bit-identical" % (c, c), 0.0, 0.0)


# the four conv geometries of SpMiddleResNetFHD (scn.py:99-143) x channel pairs from 16 to 128
CONV3D_CASES = [((3, 3, 3), (1, 1, 1), (1, 1, 1), True, 16, 16, 0.10), ((3, 3, 3), (1, 1, 1), (1, 1, 1), True, 32, 32, 0.15),
                ((3, 3, 3), (1, 1, 1), (1, 1, 1), True, 64, 64, 0.25), ((3, 3, 3), (1, 1, 1), (1, 1, 1), True, 128, 128, 0.30),
                ((3, 3, 3), (2, 2, 2), (1, 1, 1), False, 16, 32, 0.05), ((3, 3, 3), (2, 2, 2), (1, 1, 1), False, 32, 64, 0.10),
                ((3, 3, 3), (2, 2, 2), (0, 1, 1), False, 64, 128, 0.15), ((3, 1, 1), (2, 1, 1), (0, 0, 0), False, 128, 128, 0.30)]


@pytest.mark.parametrize("case", CONV3D_CASES, ids=lambda c: "k%s_s%s_p%s_%d-%d" % ("".join(map(str, c[0])), "".join(map(str, c[1])),
                                                                                       "".join(map(str, c[2])), c[4], c[5]))
def test_sparse_conv_matches_dense_conv3d_directly(hip, case):
    """An arbiter that is NOT this repo's oracle: fd_rulebook + fd_spconv_apply + fd_densify against
    torch.nn.functional.conv3d on the densified input (a 41 x 96 x 96 grid, B = 2).  spconv's definition
    (SURVEY 8a A5/A6): a strided SparseConv3d equals the dense convolution at every output site (sites without an active
    input in their window are exactly zero either way); a SubMConv3d equals it at the active input sites.  The dense
    convolution runs in float64 on the host for the narrow cases and through MIOpen fp32 on the device for the wide ones
    (an independent implementation either way)."""
    ks, st, pd, subm, cin, cout, dens = case
    rng = np.random.default_rng(cin * 131 + cout + ks[1])
    B, D, H, W = 2, 41, 96, 96
    idx, feats = _random_sparse(rng, B, D, H, W, dens, cin)
    w = (rng.standard_normal((ks[0] * ks[1] * ks[2], cin, cout)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    src = hip.SparseIndex(B, D, H, W, torch.device("cuda"))
    n_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    src.mark(_dev(idx))
    src.scan(n_dev)
    src.finalize(int(n_dev.cpu()[0]))
    x = hip.rows_permute(_dev(feats), src.lookup(_dev(idx)), cin, torch.float32, n_rows=src.n)
    if subm:
        dst = src
    else:
        dst = src.downsample(ks, st, pd)
        nd = torch.zeros(1, dtype=torch.int32, device="cuda")
        dst.scan(nd)
        dst.finalize(int(nd.cpu()[0]))
    nbr = src.rulebook(dst, ks, st, pd)
    y = hip.spconv_apply(x, hip.pack_spconv_weight(torch.from_numpy(w)).cuda(), None, nbr, dst.n, cout)
    # densify through fd_densify's general sibling: scatter rows by the index coordinates (exact, no arithmetic)
    co = dst.coords.long()
    got = torch.zeros((B, cout, dst.D, dst.H, dst.W), device="cuda")
    got[co[:, 0], :, co[:, 1], co[:, 2], co[:, 3]] = y
    dense_in = torch.zeros((B, cin, D, H, W), dtype=torch.float32)
    ii = torch.from_numpy(idx.astype(np.int64))
    dense_in[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]] = torch.from_numpy(feats)
    w5 = torch.from_numpy(w).reshape(ks[0], ks[1], ks[2], cin, cout).permute(4, 3, 0, 1, 2).contiguous()  # [Cout,Cin,kD,kH,kW]
    if cin * cout <= 32 * 32:
        ref = torch.nn.functional.conv3d(dense_in.double(), w5.double(), None, stride=st, padding=pd).float()
        how = "float64 host conv3d"
    else:
        ref = torch.nn.functional.conv3d(dense_in.cuda(), w5.cuda(), None, stride=st, padding=pd).cpu()
        how = "MIOpen fp32 conv3d"
    assert tuple(ref.shape[2:]) == (dst.D, dst.H, dst.W)
    got = got.cpu()
    if subm:
        mask = torch.zeros((B, 1, D, H, W), dtype=torch.bool)
        mask[ii[:, 0], 0, ii[:, 1], ii[:, 2], ii[:, 3]] = True
        ref = ref * mask
    else:
        active = torch.zeros((B, 1, dst.D, dst.H, dst.W), dtype=torch.bool)
        cc = co.cpu()
        active[cc[:, 0], 0, cc[:, 1], cc[:, 2], cc[:, 3]] = True
        assert float((ref * ~active).abs().max()) == 0.0, "the output set must cover every site the dense conv reaches"
    assert_close("sparse conv vs F.conv3d k%s s%s p%s %d->%d (%d rows)" % (ks, st, pd, cin, cout, src.n), got.numpy(), ref.numpy(), 1e-4, how)


def test_densify_matches_oracle(hip):
    from oracle import ops as oops

    rng = np.random.default_rng(3)
    B, D, H, W, C = 2, 2, 20, 28, 32
    idx, feats = _random_sparse(rng, B, D, H, W, 0.3, C)
    src = hip.SparseIndex(B, D, H, W, torch.device("cuda"))
    n_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    src.mark(_dev(idx))
    src.scan(n_dev)
    src.finalize(int(n_dev.cpu()[0]))
    x = hip.rows_permute(_dev(feats), src.lookup(_dev(idx)), C, torch.float32, n_rows=src.n)
    ref = oops.dense(feats, idx, B, (D, H, W)).reshape(B, C * D, H, W)
    for cl in (False, True):
        out = hip.densify(x, src, channels_last=cl)
        assert np.array_equal(out.cpu().numpy(), ref)


# ------------------------------------------------------------------------------------------------ backbone
def test_backbone_matches_reference_topology_golden(hip, golden):
    """SpMiddleResNetFHD on HIP (generic module path and fused path) vs the golden produced by the reference's
    own scn.py; 1e-3 of the output scale."""
    from futuredet_amd import build_backbone
    from futuredet_amd.synth import seeded_state_dict

    g = golden("backbone.npz")
    bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
    assert sorted(bb.state_dict().keys()) == list(g["keys"])
    bb.load_state_dict(seeded_state_dict(bb, 41), strict=False)
    bb = bb.cuda().eval()
    feats, coors = _dev(g["feats"]), _dev(g["coors"])
    grid = [int(v) for v in g["grid"]]
    with torch.no_grad():
        y_fused, ms = bb(feats, coors, 2, grid)
        y_gen, ms_gen = bb.forward_generic(feats, coors, 2, grid)
    for name, y in (("fused", y_fused), ("generic", y_gen)):
        assert_close("backbone golden (%s path)" % name, y.float().cpu().numpy(), g["y"], 1e-3)
    for k in ("conv1", "conv2", "conv3", "conv4"):
        for m in (ms, ms_gen):
            ind = m[k].indices.cpu().numpy()
            order = np.lexsort(ind.T[::-1])
            assert np.array_equal(ind[order], g["ms_%s_idx" % k])
            fs = m[k].features.float().cpu().numpy()[order].sum(1)
            assert rel_err(fs, g["ms_%s_feat_sum" % k]) <= 1e-3


# ------------------------------------------------------------------------------------------------ IoU / NMS / decode
def test_iou_matches_compiled_reference_golden(hip, golden):
    g = golden("iou.npz")
    out = hip.boxes_iou_bev(_dev(g["a"]), _dev(g["b"])).cpu().numpy()
    # device libm (sin/cos/atan2) differs from the host's in the last ulp; 1e-5 absolute on an IoU in [0,1]
    np.testing.assert_allclose(out, g["iou"], atol=2e-5, rtol=0)


def test_rotated_nms_vs_oracle(hip):
    """fd_rotated_nms and rotate_nms_pcdet vs the oracle's greedy sweep.  The two may differ only where a pair's IoU sits
    within 1e-4 of the threshold (device vs host sin/cos/atan2 differ in the last ulp): whenever the index lists differ,
    the device result must still be a greedy-NMS result under the host IoU matrix up to such pairs (every kept / dropped
    decision is checked against the kept boxes before it), and without any near-threshold pair the lists must be equal."""
    from futuredet_amd.nms import rotate_nms_pcdet
    from oracle import model as omodel
    from oracle import ops as oops

    rng = np.random.default_rng(5)
    for n in (0, 1, 63, 64, 65, 700, 1500):
        b = np.zeros((n, 7), np.float32)
        b[:, :2] = rng.uniform(-20, 20, (n, 2))
        b[:, 3] = rng.uniform(1.5, 5, n)
        b[:, 4] = rng.uniform(1, 2.5, n)
        b[:, 5] = 1.5
        b[:, 6] = rng.uniform(-3.2, 3.2, n)
        keep, cnt = hip.rotated_nms(_dev(b), 0.2)
        got = keep[: int(cnt.cpu()[0])].cpu().numpy()
        want = oops.nms(b, 0.2)
        iou = oops.boxes_iou_bev(b, b) if n else np.zeros((0, 0), np.float32)
        near = int((np.abs(np.triu(iou, 1) - 0.2) < 1e-4).sum()) if n else 0
        if not np.array_equal(got, want):
            assert near > 0, "NMS differs from the oracle although no pair is within 1e-4 of the threshold"
            bad = greedy_violations(iou, got, 0.2)
            assert not bad, "fd_rotated_nms (n=%d): %s" % (n, bad[:3])
        report("fd_rotated_nms n=%d" % n, float(len(set(got.tolist()) ^ set(want.tolist()))), 0.0, "(index differences; %d near-threshold pairs)" % near)
        if n:
            scores = torch.from_numpy(rng.random(n).astype(np.float32))
            b7 = torch.from_numpy(b)
            sel = rotate_nms_pcdet(b7.cuda(), scores.cuda(), 0.2, pre_maxsize=1000, post_max_size=83).cpu()
            ref = omodel.rotate_nms_pcdet(b7.clone(), scores.clone(), 0.2, 1000, 83)
            assert len(sel) <= 83
            if not torch.equal(sel, ref):
                # rebuild what the routine hands to the IoU kernel (box_torch_ops.py:256-262) and check every decision
                order = np.argsort(-scores.numpy(), kind="stable")[:1000]
                nb = nms_layout(b[order])
                iou2 = oops.boxes_iou_bev(nb, nb)
                assert int((np.abs(np.triu(iou2, 1) - 0.2) < 1e-4).sum()) > 0, "rotate_nms_pcdet differs without a near-threshold pair"
                pos = {int(o): i for i, o in enumerate(order)}
                kept = [pos[int(i)] for i in sel.tolist()]
                bad = greedy_violations(iou2, kept, 0.2, limit=83)
                assert not bad, "rotate_nms_pcdet (n=%d): %s" % (n, bad[:3])


TEST_CFG = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500,
                nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000, nms_post_max_size=83,
                         nms_iou_threshold=0.2),
                score_threshold=0.1, pc_range=[-54, -54], out_size_factor=8, voxel_size=[0.075, 0.075], double_flip=False)


def _match_detections(got, want, tol=1e-3):
    """Order-insensitive match of [K,9]+score+label rows; returns the number of unmatched rows on either side."""
    if len(got) == 0 or len(want) == 0:
        return len(got) + len(want)
    # 1e-3 relative to the magnitude of each component (exp(dim) and velocities can be large)
    d = (np.abs(got[:, None, :] - want[None, :, :]) / np.maximum(1.0, np.abs(want[None, :, :]))).max(-1)
    return int((d.min(1) > tol).sum() + (d.min(0) > tol).sum())


def _attribute(name, got, want, cfg=None, topk_cut=None):
    """Every detection without a counterpart must be explained by a near-threshold score or NMS pair (parity_util)."""
    from oracle import ops as oops

    cfg = cfg or TEST_CFG
    return attribute_detection_diffs(name, got, want, oops.boxes_iou_bev, cfg["score_threshold"], cfg["nms"]["nms_iou_threshold"], topk_cut=topk_cut)


def _rows(res):
    r = torch.cat([res["box3d_lidar"].float(), res["scores"][:, None].float(), res["label_preds"][:, None].float()], 1)
    return r.cpu().numpy()


@pytest.mark.parametrize("name,T,dense", [("n0", 1, False), ("n3", 7, False), ("n3dtf", 7, True), ("n0big", 1, False), ("cls", 3, False),
                                          ("rev", 7, False), ("sp", 7, False), ("wide", 7, False)])
def test_predict_matches_reference_golden(hip, golden, name, T, dense):
    """CenterHead.predict on HIP vs the reference's predict outputs (decode + rotated NMS through the compiled
    reference IoU).  Boxes within 1e-3; a detection may differ only if its score is within 1e-5 of the
    threshold or an NMS pair within 1e-4 of the IoU threshold, which the fixtures are checked not to contain
    for more than 1% of rows."""
    from futuredet_amd import build_head

    g = golden("predict.npz")
    head = build_head(dict(type="CenterHead", in_channels=64, tasks=[dict(num_class=1, class_names=["car"])],
                           dataset="nuscenes", weight=0.25, code_weights=[1.0] * 10,
                           common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                           share_conv_channel=64, dcn_head=False, timesteps=T, two_stage=False, reverse=name == "rev", sparse=name == "sp",
                           dense=dense, bev_map=False, forecast_feature=False, classify=name == "cls", wide_head=name == "wide")).cuda().eval()
    # "cls": the constructor's default mode, three-class heat-maps, channel max in the decode; "rev": the reverse mode (decoded like the
    # standard head, center_head.py:559); "sp": the sparse mode (forward + reverse task, 2 x 7 output steps, :572-587)
    ntask = T if (dense or name == "cls") else (2 if name == "sp" else 1)
    preds = [{k: _dev(g["%s_in_t%d_%s" % (name, ti, k)]) for k in ("reg", "height", "dim", "rot", "vel", "hm")}
             for ti in range(ntask)]
    B = preds[0]["hm"].shape[0]
    rets = head.predict({"metadata": [None] * B}, preds, TEST_CFG)
    for b, r in enumerate(rets):
        want = np.concatenate([g["%s_out_b%d_boxes" % (name, b)], g["%s_out_b%d_scores" % (name, b)][:, None],
                               g["%s_out_b%d_labels" % (name, b)][:, None].astype(np.float32)], 1)
        got = torch.cat([r["box3d_lidar"], r["scores"][:, None], r["label_preds"][:, None].float()], 1).cpu().numpy()
        bad = _attribute("predict golden %s b%d" % (name, b), got, want)
        if bad == 0:  # same order as the reference when nothing flipped: steps in order, score-descending inside a step
            assert_close("predict golden %s b%d rows in order" % (name, b), got, want, 1e-3)


@pytest.mark.parametrize("name,T,dense", [("circ", 7, False), ("circv", 7, False), ("circd", 7, True)])
def test_predict_circular_nms_matches_reference_golden(hip, golden, name, T, dense):
    """test_cfg.circular_nms (center_head.py:722-725 -> circle_nms_jit.py): CenterHead.predict on HIP vs the reference's own predict with
    circular_nms=True.  "circ": one radius for the standard head's shared boxes (one decode group); "circv": a radius per output step (the
    shared boxes pass the NMS once per step: a decode group per step); "circd": a task per step, each with its radius.  The predicate is
    float32 arithmetic on the box centres, which the decode reproduces bit for bit: rows equal in order."""
    from futuredet_amd import build_head

    g = golden("predict.npz")
    head = build_head(dict(type="CenterHead", in_channels=64, tasks=[dict(num_class=1, class_names=["car"])], dataset="nuscenes", weight=0.25,
                           code_weights=[1.0] * 10, common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                           share_conv_channel=64, dcn_head=False, timesteps=T, two_stage=False, dense=dense, bev_map=False, forecast_feature=False,
                           classify=False)).cuda().eval()
    preds = [{k: _dev(g["%s_in_t%d_%s" % (name, ti, k)]) for k in ("reg", "height", "dim", "rot", "vel", "hm")} for ti in range(T if dense else 1)]
    B = preds[0]["hm"].shape[0]
    cfg = dict(TEST_CFG, circular_nms=True, min_radius=[float(r) for r in g[name + "_min_radius"]])
    rets = head.predict({"metadata": [None] * B}, preds, cfg)
    for b, r in enumerate(rets):
        want = np.concatenate([g["%s_out_b%d_boxes" % (name, b)], g["%s_out_b%d_scores" % (name, b)][:, None],
                               g["%s_out_b%d_labels" % (name, b)][:, None].astype(np.float32)], 1)
        got = _rows(r)
        assert got.shape == want.shape, (name, b, got.shape, want.shape)
        assert_close("predict golden %s b%d (circular NMS) rows in order" % (name, b), got, want, 1e-3)
    # the reference's error behaviour: a scalar min_radius is not subscriptable (every shipped config writes min_radius=2), a short list runs out
    with pytest.raises(TypeError):
        head.predict({"metadata": [None] * B}, preds, dict(cfg, min_radius=2))
    with pytest.raises(IndexError):
        head.predict({"metadata": [None] * B}, preds, dict(cfg, min_radius=[2.0]))
    with pytest.raises(NotImplementedError):  # center_head.py:668-669: the branch is `pass` and predict fails on rets[0]
        head.predict({"metadata": [None] * B}, preds, dict(TEST_CFG, per_class_nms=True))


@pytest.mark.parametrize("case", ["spread", "cluster", "below_cut"])
def test_circular_nms_beyond_the_candidate_cut(hip, case):
    """The reference applies no pre-NMS cut under circular_nms; the kernels take the 4096 best candidates of a group.  "spread": 32 400
    candidates all over a 180 x 180 map -- nms_post_max_size boxes are kept long before the cut, the result is the uncut oracle's.
    "cluster": the candidates beyond the cut could still be kept (everything taken lies inside a few circles, fewer than post_max kept):
    the group reports count -1 and predict raises instead of returning a different answer.  "below_cut": 3000 candidates, all taken."""
    from futuredet_amd import build_head
    from oracle import model as omodel

    rng = np.random.default_rng(5 + len(case))
    H = W = 180
    logit = rng.standard_normal((1, 1, H, W)).astype(np.float32)
    radius = 3.0
    if case == "cluster":
        radius = 2.0e4  # one circle swallows the map: a single box is kept, 32 399 candidates lie beyond it
    elif case == "below_cut":
        logit -= 8.0
        logit.reshape(-1)[rng.choice(H * W, 3000, replace=False)] += 9.0
    preds = [dict(hm=logit, reg=rng.uniform(0, 1, (1, 2, H, W)).astype(np.float32), height=rng.normal(-1, 0.5, (1, 1, H, W)).astype(np.float32),
                  dim=rng.normal(0.5, 0.1, (1, 3, H, W)).astype(np.float32), rot=rng.standard_normal((1, 2, H, W)).astype(np.float32),
                  vel=rng.standard_normal((1, 2, H, W)).astype(np.float32))]
    kw = dict(in_channels=64, tasks=[dict(num_class=1, class_names=["car"])], common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
              share_conv_channel=64, timesteps=1, classify=False)
    head = build_head(dict(type="CenterHead", dataset="nuscenes", weight=0.25, code_weights=[1.0] * 10, **kw)).cuda().eval()
    cfg = dict(TEST_CFG, circular_nms=True, min_radius=[radius] * 7)
    dev = [{k: _dev(v) for k, v in preds[0].items()}]
    if case == "cluster":
        with pytest.raises(RuntimeError, match="circular NMS"):
            head.predict({"metadata": [None]}, dev, cfg)
        return
    got = _rows(head.predict({"metadata": [None]}, dev, cfg)[0])
    ohead = omodel.CenterHead(64, kw["tasks"], kw["common_heads"], timesteps=1, classify=False).eval()
    want = _rows(ohead.predict({"metadata": [None]}, [{k: torch.from_numpy(v) for k, v in preds[0].items()}], cfg)[0])
    assert got.shape == want.shape and len(got) == 7 * 83, (case, got.shape, want.shape)
    assert_close("circular NMS, %s: %d candidates, rows in the oracle's order" % (case, int((logit > np.log(0.1 / 0.9)).sum())), got, want, 1e-3)


@pytest.mark.parametrize("case", ["ties_across_cut", "all_equal", "pre_max_4096", "tiny_map", "few_valid"])
def test_decode_selection_edge_cases(hip, case):
    """The pre-max selection of the decode (dec_keys_hist / dec_select_hist / dec_rank_decode, fd_decode.hip) against the definition --
    the pre_max best cells by (score descending, cell ascending), box_torch_ops.py:259-261 with a stable order -- on maps built to hit
    its branches: scores tied across the cut (the threshold bin holds the ties: ranked in LDS), ALL scores equal (32k candidates in one bin:
    the radix passes), nms_pre_max_size 4096 (the round-4 kernel asked for more than 64 KB of LDS there), a map smaller than
    one block, fewer valid cells than pre_max.  NMS is switched off by an IoU threshold above 1, post_max = 128: the first 128 selected
    boxes come back in order and must be the 128 best cells."""
    rng = np.random.default_rng(len(case))
    H, W = (12, 20) if case == "tiny_map" else (180, 180)
    pre = 4096 if case == "pre_max_4096" else 100  # (100 < post_max: every selected cell comes back, so the cut itself is checked)
    logit = rng.standard_normal((1, 1, H, W)).astype(np.float32)
    if case == "ties_across_cut":
        logit = np.round(logit * 4) / 4  # ~40 distinct scores: hundreds of exact ties, the cut falls inside one of them
    elif case == "all_equal":
        logit[:] = 0.5
    elif case == "few_valid":
        logit -= 6.0
        logit.reshape(-1)[rng.choice(H * W, 60, replace=False)] += 8.0
    cfg_d = dict(TEST_CFG, score_threshold=0.1, nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=pre, nms_post_max_size=128,
                                                         nms_iou_threshold=1.5))
    cfg = hip.make_decode_cfg(H, W, cfg_d)
    reg = rng.uniform(0, 1, (1, 2, H, W)).astype(np.float32)
    hei = rng.normal(-1, 0.5, (1, 1, H, W)).astype(np.float32)
    dim = rng.normal(0.5, 0.1, (1, 3, H, W)).astype(np.float32)
    rot = rng.standard_normal((1, 2, H, W)).astype(np.float32)
    boxes, scores, cell, count = hip.centerpoint_decode(_dev(logit), _dev(reg), _dev(hei), _dev(dim), _dev(rot), cfg)
    score = (1.0 / (1.0 + np.exp(-logit.astype(np.float32)))).astype(np.float32).reshape(-1)
    ys, xs = np.divmod(np.arange(H * W), W)
    cx = ((xs + reg[0, 0].reshape(-1)) * np.float32(8) * np.float32(0.075) + np.float32(-54)).astype(np.float32)
    cy = ((ys + reg[0, 1].reshape(-1)) * np.float32(8) * np.float32(0.075) + np.float32(-54)).astype(np.float32)
    z = hei.reshape(-1)
    lim = cfg_d["post_center_limit_range"]
    ok = (score > 0.1) & (cx >= lim[0]) & (cy >= lim[1]) & (z >= lim[2]) & (cx <= lim[3]) & (cy <= lim[4]) & (z <= lim[5])
    # (equal logits give equal scores on either side, distinct logits differ by far more than an ulp of expf: the order is that of the logits)
    order = np.lexsort((np.arange(H * W), -logit.reshape(-1).astype(np.float64)))  # score descending, cell ascending
    order = order[ok[order]][:pre]
    n = int(count.cpu()[0])
    assert n == min(128, len(order)), (n, len(order))
    got_cell, got_score = cell.cpu().numpy()[0, :n], scores.cpu().numpy()[0, :n]
    assert np.allclose(got_score, score[order[:n]], rtol=0, atol=2e-7)
    assert np.array_equal(got_cell, order[:n]), case
    report("decode selection edge case %s: %d selected, first %d in the reference order" % (case, len(order), n), 0.0, 0.0)


def test_bf16_sweeps_in_flight_are_deterministic(hip):
    """Four captured bf16 sweeps in flight on four streams (bench.py's shape), 120 rounds: every replay of a stream returns its first
    replay's packed detections bit for bit.  Round 6 found this failing -- 0.5-3 % of the sweeps came back with another detection list:
    rotated-IoU decisions flipped in nms_mask while its inputs were bit-identical, only in lanes 48-63, only next to the bf16 dense
    convolution of another stream, only when the IoU geometry was compiled to packed-fp32 instructions (fd_decode.hip, the comment on
    footprint_overlap; futuredet_amd/build.py: -fno-slp-vectorize; profiles/round6_determinism_soak.txt).  480 replays caught the old
    build in every run of the soak (expected 2-14 differing replays)."""
    from futuredet_amd import build_detector
    from futuredet_amd.configs import centerpoint_config
    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims

    cfg = centerpoint_config("forecast_n3")
    net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    net.load_state_dict(tame_box_dims(seeded_state_dict(net, 7)), strict=False)
    net = net.cuda().eval()
    net.set_precision(torch.bfloat16)
    B, NS, rounds = 2, 4, 120
    clouds = [[torch.from_numpy(synthetic_cloud(seed=10 * s + b, target_points=300000)).cuda() for b in range(B)] for s in range(NS)]
    cap = max(c.shape[0] for cs in clouds for c in cs) + 1024
    streams = [torch.cuda.Stream() for _ in range(NS)]
    steps = []
    with torch.no_grad():
        for s, st in enumerate(streams):
            with torch.cuda.stream(st):
                step = StaticStep(net, cfg.voxel_generator, cap, batch_size=B, ndim=5, packed=True, row_caps="auto")
                step.warm_up(clouds[s])
                step.capture()
                steps.append(step)
        torch.cuda.synchronize()
        first, differing = [None] * NS, 0
        for r in range(rounds):
            snaps = []
            for s, st in enumerate(streams):
                with torch.cuda.stream(st):
                    packed, counts = steps[s](clouds[s], check=False)
                    snaps.append((packed.clone(), counts.clone(), steps[s].level_counts.clone()))
            torch.cuda.synchronize()
            for s, snap in enumerate(snaps):
                assert not steps[s].overflowed(snap[2].cpu().tolist())
                if first[s] is None:
                    first[s] = snap
                    assert int(snap[1].sum()) > 500
                else:
                    differing += not (torch.equal(snap[0], first[s][0]) and torch.equal(snap[1], first[s][1]))
    report("bf16 sweeps, %d in flight x %d rounds: replays that differ from the stream's first" % (NS, rounds), float(differing), 0.0)
    assert differing == 0


# ------------------------------------------------------------------------------------------------ end to end
@pytest.mark.parametrize("variant", ["forecast_n0", "forecast_n3", "pedestrian_n3_fine", "forecast_n3dtfm"])
def test_voxelnet_end_to_end_vs_oracle(hip, variant):
    """Whole path on a ~30k-point synthetic cloud (BASELINE configs[0] shape): HIP VoxelNet.forward(example) and
    forward_points() vs the CPU oracle model with the same seeded weights.  BEV map element-wise
    |d| <= 1e-3 * max(1, |ref|); detections matched within 1e-3 per component, every unmatched one attributed to a
    near-threshold score / IoU pair (parity_util.attribute_detection_diffs)."""
    from futuredet_amd import build_detector
    from futuredet_amd.collate import collate_kitti_multi, example_to_device
    from futuredet_amd.configs import centerpoint_config
    from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims
    from futuredet_amd.voxelize import Voxelization
    from oracle import model as omodel
    from oracle import ops as oops

    if variant == "pedestrian_n3_fine":  # BASELINE configs[4]: pedestrian n3 on the finer 0.05 m grid (2160^2, BEV 270^2)
        cfg = centerpoint_config("forecast_n3", "pedestrian", voxel_size=(0.05, 0.05, 0.2), max_voxel_num=(300000, 400000))
    else:
        cfg = centerpoint_config(variant)
    net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = tame_box_dims(seeded_state_dict(net, 7))
    net.load_state_dict(sd, strict=False)
    net = net.cuda().eval()
    onet = omodel.VoxelNet(cfg.model["reader"], cfg.model["backbone"], cfg.model["neck"], cfg.model["bbox_head"],
                           test_cfg=cfg.test_cfg).eval()
    missing, unexpected = onet.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    clouds = [synthetic_cloud(seed=s, target_points=30000) for s in (0, 1)]
    vox = Voxelization(cfg=cfg.voxel_generator)
    examples, oexamples = [], []
    for i, pts in enumerate(clouds):
        res, _ = vox({"mode": "val", "lidar": {"points": pts}}, None)
        v = res["lidar"]["voxels"]
        ov, oc, on = oops.points_to_voxel(pts, cfg.voxel_generator["voxel_size"], cfg.voxel_generator["range"], 10, True,
                                          cfg.voxel_generator["max_voxel_num"][1])
        assert np.array_equal(v["coordinates"], oc) and np.array_equal(v["voxels"], ov)
        examples.append(dict(voxels=v["voxels"], coordinates=v["coordinates"], num_points=v["num_points"],
                             num_voxels=v["num_voxels"], shape=v["shape"], metadata={"token": i}))
        if cfg.BEV_MAP:  # n3dtfm: 6-channel rasterised map input of the head's bev_conv branch (center_head.py:336-341,380-381)
            examples[-1]["bev_map"] = np.random.default_rng(90 + i).uniform(0, 1, (6, 180, 180)).astype(np.float32)
    batch = collate_kitti_multi(examples)
    with torch.no_grad():
        want = onet(batch)
        obev = onet.extract_feat(batch)
        dev_batch = example_to_device(batch, torch.device("cuda"))
        got = net(dev_batch, return_loss=False)
        x, _ = net.extract_feat(dict(features=dev_batch["voxels"], num_voxels=dev_batch["num_points"],
                                     coors=dev_batch["coordinates"], batch_size=2, input_shape=dev_batch["shape"][0]))
        bev_in = torch.stack(dev_batch["bev_map"], dim=1).float() if cfg.BEV_MAP else None
        fast = net.forward_points([_dev(c) for c in clouds], cfg.voxel_generator, bev_map=bev_in, padded=False)
    assert_close("e2e 30k %s BEV map (neck output)" % variant, x.float().cpu().numpy(), obev.numpy(), 1e-3)
    for b in range(2):
        w = _rows(want[b])
        for tag, res in (("forward", got), ("forward_points", fast)):
            gt = _rows(res[b])
            if cfg.DENSE:
                # seven chained task heads with random weights saturate many logits to a score of exactly 1.0; the order of
                # such ties (and hence top-k / NMS membership, and what those boxes suppress) is not defined, so only
                # unsaturated detections are matched and the tie cascade is bounded by count instead of attributed
                gt, wu = gt[gt[:, 9] < 0.999], w[w[:, 9] < 0.999]
                bad = _match_detections(gt, wu)
                report("e2e 30k %s %s b%d detections" % (variant, tag, b), float(bad), 0.02 * (len(gt) + len(wu)), "(unsaturated rows)")
                assert bad <= max(2, 0.02 * (len(gt) + len(wu))), (variant, b, bad, len(gt), len(wu))
            else:
                _attribute("e2e 30k %s %s b%d" % (variant, tag, b), gt, w, cfg.test_cfg)
    if cfg.DENSE:  # and the raw head outputs of the chain agree tensor by tensor
        with torch.no_grad():
            op = onet.bbox_head(obev, torch.stack(batch["bev_map"], dim=1).float())
            hp = net.bbox_head(x, bev_in)
        for t in (0, 3, 6):
            for k in op[t]:
                assert_close("e2e 30k %s head t%d %s" % (variant, t, k), hp[t][k].float().cpu().numpy(), op[t][k].numpy(), 1e-3)


# ------------------------------------------------------------------------------------------------ dense conv (bf16)
@pytest.mark.parametrize("cfg", [(3, 1, 64, 128, 37, 45), (3, 2, 32, 128, 40, 33), (3, 1, 128, 64, 20, 20), (3, 1, 96, 11, 19, 35),
                                 (1, 1, 128, 256, 23, 18), (3, 1, 256, 256, 16, 16)])
def test_conv2d_nhwc_bf16_vs_torch(hip, cfg):
    """Hand-written MFMA conv vs torch conv2d on the same bf16-rounded operands (fp32 reference); partial tiles,
    stride 2, 1x1, padded Cout, channel-offset (concat) writes.  bf16 output rounding -> 1e-2 of scale."""
    ks, stride, cin, cout, H, W = cfg
    rng = np.random.default_rng(cin + cout + H)
    x = torch.from_numpy(rng.standard_normal((2, cin, H, W)).astype(np.float32)).bfloat16()
    w = torch.from_numpy((rng.standard_normal((cout, cin, ks, ks)) * (2.0 / (cin * ks * ks)) ** 0.5).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    ref = torch.relu(torch.nn.functional.conv2d(x.float(), w.bfloat16().float(), b, stride=stride, padding=1 if ks == 3 else 0))
    wpk = hip.pack_conv2d_weight(w).cuda()
    xn = x.cuda().permute(0, 2, 3, 1).contiguous()
    out = torch.full((2, ref.shape[2], ref.shape[3], cout + 5), 7.0, dtype=torch.bfloat16, device="cuda")
    hip.conv2d_nhwc_bf16(xn, wpk, b.cuda(), cout, ks, stride, True, out=out, co_off=3)
    got = out[..., 3:3 + cout].permute(0, 3, 1, 2).float().cpu()
    assert float((got - ref).abs().max()) <= 1e-2 * max(1.0, float(ref.abs().max()))
    assert bool((out[..., :3] == 7).all()) and bool((out[..., 3 + cout:] == 7).all()), "writes outside the channel window"


STRIP_CASES = [(3, 64, 128, 1, 180, 180), (3, 128, 64, 2, 90, 90), (3, 32, 32, 1, 66, 64), (3, 96, 11, 2, 5, 127), (3, 64, 256, 1, 17, 129),
               (1, 128, 256, 2, 23, 70), (1, 64, 40, 1, 9, 300), (3, 32, 128, 1, 1, 128), (3, 64, 64, 3, 2, 65), (3, 32, 70, 1, 40, 270),
               # narrower than a strip (tile kernels only): maps smaller than one 8 x 16 tile in either direction, exact multiples, one-pixel edges
               (3, 32, 32, 1, 40, 9), (3, 32, 64, 2, 3, 5), (1, 64, 64, 1, 16, 32), (3, 32, 128, 1, 15, 31), (3, 64, 32, 1, 33, 47), (3, 32, 32, 1, 9, 17)]


@pytest.mark.parametrize("cfg", STRIP_CASES, ids=lambda c: "k%d_%d-%d_b%d_%dx%d" % c)
def test_conv2d_bf16_strip_kernel_matches_tile_kernel_and_torch(hip, cfg):
    """Stride-1 bf16 layers on images at least 64 pixels wide can run on strips of 128 consecutive pixels (one workgroup per compute
    unit at 180 x 180 instead of 276 tiles for 256 units; opt-in, conv_strip = 1: faster alone, slower with sweeps in flight).  Same
    summation order as the 8 x 16-tile kernel: the two must agree BIT FOR BIT (conv_strip = 0, the default, selects the tile kernel on its
    MIXED tiling -- whole 8 x 16 tiles, then tiles of other shapes for the columns right of them and the rows below: 254 workgroups
    instead of 276 at 180 x 180 -- and -1 the ragged 8 x 16 grid; all three are compared) for every channel block width, on shapes whose strips touch one, two
    and three image rows, end in a partial strip, cross no image boundary in a batch, with a channel-offset output window and
    the pixel-shuffle placement of the 2 x 2 transposed convolution; and both match torch on the bf16-rounded operands."""
    ks, cin, cout, B, H, W = cfg
    rng = np.random.default_rng(cin + cout + H + W)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32)).bfloat16()
    w = torch.from_numpy((rng.standard_normal((cout, cin, ks, ks)) * (2.0 / (cin * ks * ks)) ** 0.5).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    ref = torch.relu(torch.nn.functional.conv2d(x.float(), w.bfloat16().float(), b, padding=1 if ks == 3 else 0))
    wpk = hip.pack_conv2d_weight(w).cuda()
    xn = x.cuda().permute(0, 2, 3, 1).contiguous()
    outs = {}
    try:
        for strip in (1, 0, -1):  # strips / the default (mixed tiling: whole 8 x 16 tiles + edge tiles of other shapes) / ragged 8 x 16 tiles only
            hip.set_tuning("conv_strip", strip)
            for nt in (0, 64, 32):
                hip.set_tuning("conv_nt", nt)
                out = torch.full((B, H, W, cout + 8), 7.0, dtype=torch.bfloat16, device="cuda")
                hip.conv2d_nhwc_bf16(xn, wpk, b.cuda(), cout, ks, 1, True, out=out, co_off=8 if cout % 8 == 0 else 3)
                outs[(strip, nt)] = out
                # pixel-shuffle placement (the transposed convolution's sub-convolutions): output pixel (2 y + 1, 2 x)
                sh = torch.full((B, 2 * H, 2 * W, cout), 5.0, dtype=torch.bfloat16, device="cuda")
                hip.conv2d_nhwc_bf16(xn, wpk, b.cuda(), cout, ks, 1, False, out=sh, osy=2, osx=2, ooy=1, oox=0)
                outs[(strip, nt, "shuffle")] = sh
    finally:
        hip.set_tuning("conv_strip", 0)
        hip.set_tuning("conv_nt", 0)
    base = outs[(-1, 0)]
    off = 8 if cout % 8 == 0 else 3
    got = base[..., off:off + cout].permute(0, 3, 1, 2).float().cpu()
    assert float((got - ref).abs().max()) <= 1e-2 * max(1.0, float(ref.abs().max()))
    for key, o in outs.items():
        ref_o = outs[(-1, key[1], "shuffle")] if len(key) == 3 else outs[(-1, key[1])]
        assert torch.equal(o, ref_o), "strip kernel differs from the tile kernel: %s" % (key,)
        if len(key) == 2:
            assert torch.equal(o, base), "channel block width changes the result: %s" % (key,)
            assert bool((o[..., :off] == 7).all()) and bool((o[..., off + cout:] == 7).all()), "writes outside the channel window"
        else:
            assert bool((o[:, 0::2] == 5).all()) and bool((o[:, 1::2, 1::2] == 5).all()), "writes outside the shuffled pixels"


F32_CONV_CASES = [(3, 1, 64, 128, 37, 45, 0), (3, 2, 32, 128, 40, 33, 0), (3, 1, 128, 64, 20, 20, 0), (3, 1, 96, 11, 19, 35, 0),
                  (1, 1, 128, 256, 23, 18, 0), (3, 1, 256, 256, 16, 16, 0), (3, 1, 384, 23, 30, 26, 0)] + \
                 [(3, 1, 48, 70, 29, 31, k) for k in range(1, 17)] + [(3, 2, 16, 130, 33, 27, k) for k in (1, 3, 9, 12, 14)] + \
                 [(1, 1, 80, 40, 21, 22, k) for k in (2, 10, 13, 17, 18, 19, 20, 21, 22, 23, 24)] + [(1, 1, 48, 300, 9, 7, k) for k in (0, 17, 20, 21, 22, 23, 24)]


@pytest.mark.parametrize("cfg", F32_CONV_CASES, ids=lambda c: "k%ds%d_%d-%d_%dx%d_t%d" % c)
def test_conv2d_nhwc_f32_vs_torch(hip, cfg):
    """Hand-written fp32 MFMA conv vs torch conv2d in float64 on the host: partial tiles, stride 2, 1x1, Cout that is not
    a multiple of 16 / 64, channel-offset (concat) writes, every tile shape of the dispatcher (tile = 1..16 selects one,
    17..24 the pointwise GEMM variants of a 1x1 convolution; 0 = the library heuristic).  fp32 FMA chains of <= 9 * 384 terms: |d| <= 1e-4 * max(1, |ref|)."""
    ks, stride, cin, cout, H, W, tile = cfg
    rng = np.random.default_rng(cin + cout + H + tile)
    x = torch.from_numpy(rng.standard_normal((2, cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((cout, cin, ks, ks)) * (2.0 / (cin * ks * ks)) ** 0.5).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1 if ks == 3 else 0)).float()
    wpk = hip.pack_conv2d_weight_f32(w).cuda()
    xn = x.cuda().permute(0, 2, 3, 1).contiguous()
    out = torch.full((2, ref.shape[2], ref.shape[3], cout + 5), 7.0, dtype=torch.float32, device="cuda")
    assert hip.conv2d_f32_num_tiles() == 24  # 16 direct-kernel tiles + 8 pointwise (1x1 only) variants (21, 22: pixels shared through LDS per slice; 23, 24: whole pixel rows staged once)
    hip.conv2d_nhwc_f32(xn, wpk, b.cuda(), cout, ks, stride, True, out=out, co_off=3, tile=tile)
    got = out[..., 3:3 + cout].permute(0, 3, 1, 2).cpu()
    assert_close("conv2d_nhwc_f32 k%d s%d %d->%d %dx%d tile %d" % cfg, got.numpy(), ref.numpy(), 1e-4)
    assert bool((out[..., :3] == 7).all()) and bool((out[..., 3 + cout:] == 7).all()), "writes outside the channel window"


@pytest.mark.parametrize("cfg", [(64, 128, 37, 45, 1), (128, 128, 20, 20, 2), (96, 11, 19, 35, 3), (256, 256, 16, 16, 4), (48, 70, 29, 31, 0),
                                 (16, 384, 8, 50, 1), (32, 64, 33, 6, 4), (128, 128, 21, 20, 5), (64, 70, 18, 23, 6),
                                 (64, 128, 37, 77, 7), (96, 11, 19, 67, 7), (256, 256, 9, 63, 7), (48, 70, 29, 90, 7), (16, 384, 8, 65, 7),
                                 (128, 128, 45, 180, 7)], ids=lambda c: "%d-%d_%dx%d_t%d" % c)
def test_conv2d_wino_f32_vs_torch(hip, cfg):
    """Winograd F(2x2,3x3) on MFMA vs torch conv2d in float64: odd sizes (partial 2x2 tiles and partial workgroup tiles), Cout
    not a multiple of 16 / 64, channel-offset writes, every workgroup tile (tile 7 = strips of 32 tiles that wrap to the next
    tile row, cross into the next image of the batch and end beyond the last tile).  The transforms add a few roundings to an fp32
    chain of <= 256 terms per product: |d| <= 2e-4 * max(1, |ref|)."""
    cin, cout, H, W, tile = cfg
    rng = np.random.default_rng(cin + cout + H + tile)
    x = torch.from_numpy(rng.standard_normal((2, cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (cin * 9)) ** 0.5).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)).float()
    wpk = hip.pack_conv2d_weight_wino(w).cuda()
    xn = x.cuda().permute(0, 2, 3, 1).contiguous()
    out = torch.full((2, H, W, cout + 5), 7.0, dtype=torch.float32, device="cuda")
    hip.conv2d_wino_nhwc_f32(xn, wpk, b.cuda(), cout, True, out=out, co_off=3, tile=tile)
    got = out[..., 3:3 + cout].permute(0, 3, 1, 2).cpu()
    assert_close("conv2d_wino_f32 %d->%d %dx%d tile %d" % cfg, got.numpy(), ref.numpy(), 2e-4)
    assert bool((out[..., :3] == 7).all()) and bool((out[..., 3 + cout:] == 7).all()), "writes outside the channel window"


@pytest.mark.parametrize("tile", [0, 17, 18, 19, 20, 21, 22, 23, 24])
def test_conv1x1_f32_row_contiguous_epilogue_vs_torch(hip, tile):
    """The pointwise kernels' row-contiguous epilogue (round 6: accumulators leave through a per-wave LDS tile as 16-byte pieces of a pixel's
    channel run) and the LDS-shared pixel operand (tiles 21 / 22) vs torch in float64: channel counts whose 64-channel wave blocks are all real
    (256, 192) and one that mixes both epilogues (80: the last block is partial), into an aligned channel window of a wider tensor, partial pixel
    blocks; and ConvTranspose2d(2, stride 2) as one 1x1 convolution + pixel shuffle with 128-channel sub-convolutions."""
    rng = np.random.default_rng(tile)
    for cin, cout, H, W in ((128, 256, 23, 18), (80, 192, 9, 31), (48, 80, 21, 22)):
        x = torch.from_numpy(rng.standard_normal((2, cin, H, W)).astype(np.float32))
        w = torch.from_numpy((rng.standard_normal((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5).astype(np.float32))
        b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
        ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double())).float()
        out = torch.full((2, H, W, cout + 16), 7.0, dtype=torch.float32, device="cuda")
        hip.conv2d_nhwc_f32(x.cuda().permute(0, 2, 3, 1).contiguous(), hip.pack_conv2d_weight_f32(w).cuda(), b.cuda(), cout, 1, 1, True, out=out, co_off=8, tile=tile)
        assert_close("conv1x1_f32 %d->%d %dx%d tile %d (aligned window)" % (cin, cout, H, W, tile), out[..., 8:8 + cout].permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), 1e-4)
        assert bool((out[..., :8] == 7).all()) and bool((out[..., 8 + cout:] == 7).all()), "writes outside the channel window"
    cin, cout, k, H, W = 64, 128, 2, 13, 21
    x = torch.from_numpy(rng.standard_normal((2, cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((cin, cout, k, k)) * 0.1).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    ref = torch.relu(torch.nn.functional.conv_transpose2d(x.double(), w.double(), b.double(), stride=k)).float()
    wv = w.permute(2, 3, 1, 0).reshape(-1, cin)[:, :, None, None].contiguous()
    out = torch.full((2, H * k, W * k, cout + 8), 7.0, device="cuda")
    hip.conv2d_shuffle_nhwc_f32(x.cuda().permute(0, 2, 3, 1).contiguous(), hip.pack_conv2d_weight_f32(wv).cuda(), b.cuda(), cout, k, True, out=out, co_off=4,
                                tile=tile)
    assert_close("conv2d_shuffle_f32 %d->%d k%d tile %d" % (cin, cout, k, tile), out[..., 4:4 + cout].permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), 1e-4)
    assert bool((out[..., :4] == 7).all()) and bool((out[..., 4 + cout:] == 7).all())


def test_conv2d_shuffle_and_grouped_f32_vs_torch(hip):
    """fd_conv2d_shuffle_nhwc_f32 (ConvTranspose2d(k, stride k) as one 1x1 conv + pixel shuffle, into a channel slice) and
    fd_conv2d_grouped_nhwc_f32 (the CenterHead branches' final convs in one launch) vs torch in float64, 1e-4 element-wise."""
    rng = np.random.default_rng(11)
    for (cin, cout, k, H, W) in ((64, 32, 2, 13, 21), (32, 20, 3, 9, 7)):
        x = torch.from_numpy(rng.standard_normal((2, cin, H, W)).astype(np.float32))
        w = torch.from_numpy((rng.standard_normal((cin, cout, k, k)) * 0.1).astype(np.float32))
        b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
        ref = torch.relu(torch.nn.functional.conv_transpose2d(x.double(), w.double(), b.double(), stride=k)).float()
        wv = w.permute(2, 3, 1, 0).reshape(-1, cin)[:, :, None, None].contiguous()
        out = torch.full((2, H * k, W * k, cout + 8), 7.0, device="cuda")
        hip.conv2d_shuffle_nhwc_f32(x.cuda().permute(0, 2, 3, 1).contiguous(), hip.pack_conv2d_weight_f32(wv).cuda(), b.cuda(), cout, k, True,
                                    out=out, co_off=4)
        assert_close("conv2d_shuffle_f32 %d->%d k%d" % (cin, cout, k), out[..., 4:4 + cout].permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), 1e-4)
        assert bool((out[..., :4] == 7).all()) and bool((out[..., 4 + cout:] == 7).all())
    counts, cin_g = [2, 1, 3, 2, 14, 1], 64
    H, W = 19, 27
    x = torch.from_numpy(rng.standard_normal((2, cin_g * len(counts), H, W)).astype(np.float32))
    ws = [torch.from_numpy((rng.standard_normal((c, cin_g, 3, 3)) * 0.05).astype(np.float32)) for c in counts]
    bs = [torch.from_numpy(rng.standard_normal(c).astype(np.float32)) for c in counts]
    ref = torch.cat([torch.nn.functional.conv2d(x[:, g * cin_g:(g + 1) * cin_g].double(), ws[g].double(), bs[g].double(), padding=1)
                     for g in range(len(counts))], 1).float()
    wg = torch.zeros((16 * len(counts), cin_g, 3, 3))
    bg = torch.zeros((16 * len(counts),))
    for g, c in enumerate(counts):
        wg[16 * g:16 * g + c], bg[16 * g:16 * g + c] = ws[g], bs[g]
    for tile in (0, 12):
        out = hip.conv2d_grouped_nhwc_f32(x.cuda().permute(0, 2, 3, 1).contiguous(), hip.pack_conv2d_weight_f32(wg).cuda(), bg.cuda(), counts, cin_g, tile=tile)
        assert tuple(out.shape) == (2, H, W, sum(counts))
        assert_close("conv2d_grouped_f32 6 x 64 -> %s tile %d" % (counts, tile), out.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), 1e-4)


def test_dense_f32_plan_vs_torch_modules(hip):
    """RPN + CenterHead (n3: 7-timestep head) on the fp32 MFMA plan (concat in place, transposed conv as 4 interleaved
    1x1, fused heads) vs the plain fp32 torch modules (MIOpen): |d| <= 1e-3 * max(1, |ref|) per element."""
    import logging

    from futuredet_amd import build_head, build_neck
    from futuredet_amd.synth import seeded_state_dict

    rpn = build_neck(dict(type="RPN", layer_nums=[2, 2], ds_layer_strides=[1, 2], ds_num_filters=[64, 128], us_layer_strides=[1, 2],
                          us_num_filters=[128, 128], num_input_features=64, logger=logging.getLogger("RPN")))
    rpn.load_state_dict(seeded_state_dict(rpn, 3), strict=False)
    head = build_head(dict(type="CenterHead", in_channels=256, tasks=[dict(num_class=1, class_names=["car"])], dataset="nuscenes",
                           weight=0.25, code_weights=[1.0] * 10,
                           common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                           share_conv_channel=64, dcn_head=False, timesteps=7, two_stage=False, reverse=False, sparse=False, dense=False,
                           bev_map=False, forecast_feature=False, classify=False, wide_head=False))
    head.load_state_dict(seeded_state_dict(head, 4), strict=False)
    rpn, head = rpn.cuda().eval(), head.cuda().eval()
    x = torch.randn((2, 64, 44, 36), device="cuda")
    with torch.no_grad():
        y_ref = rpn.forward_modules(x)
        p_ref = head.forward_modules(y_ref)
        y = rpn(x)
        p = head(y)
        assert rpn._plan[1] is not None and head._plan[1] is not None, "the fp32 plan must be the path that ran"
    assert_close("fp32 plan RPN vs torch modules", y.cpu().numpy(), y_ref.cpu().numpy(), 1e-3)
    for k in p_ref[0]:
        assert p[0][k].shape == p_ref[0][k].shape
        assert_close("fp32 plan head %s vs torch modules" % k, p[0][k].float().cpu().numpy(), p_ref[0][k].cpu().numpy(), 1e-3)


def test_dense_bf16_plan_vs_torch_modules(hip):
    """RPN + CenterHead on the HIP bf16 plan (concat in place, transposed conv as 4 interleaved 1x1, fused heads) vs
    the plain fp32 torch modules; bf16 tolerance 3e-2 of each tensor's scale."""
    import logging

    from futuredet_amd import build_head, build_neck
    from futuredet_amd.synth import seeded_state_dict

    rpn = build_neck(dict(type="RPN", layer_nums=[2, 2], ds_layer_strides=[1, 2], ds_num_filters=[64, 128], us_layer_strides=[1, 2],
                          us_num_filters=[128, 128], num_input_features=64, logger=logging.getLogger("RPN")))
    rpn.load_state_dict(seeded_state_dict(rpn, 3), strict=False)
    head = build_head(dict(type="CenterHead", in_channels=256, tasks=[dict(num_class=1, class_names=["car"])], dataset="nuscenes",
                           weight=0.25, code_weights=[1.0] * 10,
                           common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                           share_conv_channel=64, dcn_head=False, timesteps=7, two_stage=False, reverse=False, sparse=False, dense=False,
                           bev_map=False, forecast_feature=False, classify=False, wide_head=False))
    head.load_state_dict(seeded_state_dict(head, 4), strict=False)
    rpn, head = rpn.cuda().eval(), head.cuda().eval()
    x = torch.randn((2, 64, 44, 36), device="cuda")
    with torch.no_grad():
        y_ref = rpn.forward_modules(x)
        p_ref = head.forward_modules(y_ref)
        rpn.compute_dtype = head.compute_dtype = torch.bfloat16
        y = rpn(x)
        p = head(y)
    assert y.shape == y_ref.shape
    assert float((y.float() - y_ref).abs().max()) <= 3e-2 * float(y_ref.abs().max())
    for k in p_ref[0]:
        assert p[0][k].shape == p_ref[0][k].shape
        assert float((p[0][k] - p_ref[0][k]).abs().max()) <= 3e-2 * max(1.0, float(p_ref[0][k].abs().max())), k


def test_default_classify_head_on_the_plan_and_packed_decode(hip, golden):
    """CenterHead built WITHOUT the ``classify`` keyword (the reference's default is True, center_head.py:253): one task per timestep
    with a three-class heat-map.  The device path (convolution plan, fp32) reproduces the reference's forward (dense_nets.npz,
    "cls3"), and the packed decode -- score = maximum over the three heat-map channels, fd_decode_cfg.hm_channels -- returns the
    detections of the per-task path."""
    from futuredet_amd import build_head
    from futuredet_amd.synth import seeded_state_dict

    g = golden("dense_nets.npz")
    head = build_head(dict(type="CenterHead", in_channels=64, tasks=[dict(num_class=1, class_names=["car"])], dataset="nuscenes", weight=0.25,
                           code_weights=[1.0] * 10, common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                           share_conv_channel=64, timesteps=3))
    assert head.classify and not head.standard and len(head.tasks) == 3
    head.load_state_dict(seeded_state_dict(head, 12), strict=False)
    head = head.cuda().eval()
    y = _dev(g["rpn_y"])
    with torch.no_grad():
        preds = head(y)
    assert head._plan[1] is not None and preds[0].raw is not None, "the convolution plan must be the path that ran"
    for ti, pd in enumerate(preds):
        assert pd["hm"].shape[1] == 3
        for k, v in pd.items():
            assert_close("classify head on the plan, task %d %s vs reference golden" % (ti, k), v.float().cpu().numpy(), g["head_cls3_t%d_%s" % (ti, k)], 1e-3)
    cfg = dict(TEST_CFG, score_threshold=0.01)
    packed = head.predict({"metadata": [None] * y.shape[0]}, preds, cfg)                    # decode reads the plan's NHWC buffer, channel max in dec_keys
    plain = head.predict({"metadata": [None] * y.shape[0]}, [dict(pd) for pd in preds], cfg)  # per-task tensors (raw is lost with the dict copy)
    for b in range(y.shape[0]):
        assert len(packed[b]["scores"]) > 0
        assert np.array_equal(_rows(packed[b]), _rows(plain[b]))
        assert set(packed[b]["label_preds"].tolist()) <= {0, 1, 2}


@pytest.mark.parametrize("mode,T", [("reverse", 3), ("sparse", 7), ("wide_head", 7)])
def test_reverse_and_sparse_heads_on_the_plan_and_packed_decode(hip, golden, mode, T):
    """CenterHead's ``reverse`` and ``sparse`` modes (center_head.py:322-324,559,572-587; no shipped config turns them on): the device path
    (convolution plan, fp32) reproduces the reference's forward (dense_nets.npz "rev3" / "sp7": one task / a forward and a reverse task, a
    velocity pair per timestep each) and the packed decode returns the detections of the per-task path -- 2 x 7 output steps for ``sparse``,
    the forward task's first."""
    from futuredet_amd import build_head
    from futuredet_amd.synth import seeded_state_dict

    g = golden("dense_nets.npz")
    name = {"reverse": "rev3", "sparse": "sp7", "wide_head": "wide7"}[mode]
    head = build_head(dict(type="CenterHead", in_channels=64, tasks=[dict(num_class=1, class_names=["car"])], dataset="nuscenes", weight=0.25,
                           code_weights=[1.0] * 10, common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                           share_conv_channel=64, timesteps=T, classify=False, reverse=mode == "reverse", sparse=mode == "sparse", wide_head=mode == "wide_head"))
    assert not head.standard and len(head.tasks) == (2 if mode == "sparse" else 1)
    head.load_state_dict(seeded_state_dict(head, 12), strict=False)
    head = head.cuda().eval()
    y = _dev(g["rpn_y"])
    with torch.no_grad():
        preds = head(y)
    assert head._plan[1] is not None and preds[0].raw is not None, "the convolution plan must be the path that ran"
    for ti, pd in enumerate(preds):
        assert pd["vel"].shape[1] == (2 if mode == "wide_head" else 2 * T)
        for k, v in pd.items():
            assert_close("%s head on the plan, task %d %s vs reference golden" % (mode, ti, k), v.float().cpu().numpy(), g["head_%s_t%d_%s" % (name, ti, k)], 1e-3)
    cfg = dict(TEST_CFG, score_threshold=0.01)
    packed = head.predict({"metadata": [None] * y.shape[0]}, preds, cfg)
    plain = head.predict({"metadata": [None] * y.shape[0]}, [dict(pd) for pd in preds], cfg)
    steps = 2 * T if mode == "sparse" else T
    for b in range(y.shape[0]):
        assert len(packed[b]["scores"]) > 0
        assert np.array_equal(_rows(packed[b]), _rows(plain[b]))
        assert set(packed[b]["label_preds"].tolist()) <= set(range(steps))
    # circular NMS through the packed decode (one launch set from the plan's NHWC buffer) = through the per-group path; a radius per output
    # step: equal inside every decode group -> packed, different -> the head falls back to a decode group per step (same results either way)
    for radii in ([1.5] * steps, [0.5 + 0.5 * s for s in range(steps)]):
        ccfg = dict(cfg, circular_nms=True, min_radius=radii)
        packed = head.predict({"metadata": [None] * y.shape[0]}, preds, ccfg)
        plain = head.predict({"metadata": [None] * y.shape[0]}, [dict(pd) for pd in preds], ccfg)
        for b in range(y.shape[0]):
            assert len(packed[b]["scores"]) > 0
            assert np.array_equal(_rows(packed[b]), _rows(plain[b]))
    assert head.predict_packed(preds, dict(cfg, circular_nms=True, min_radius=[1.5] * steps)) is not None or mode == "wide_head"


def test_unsupported_dense_stack_raises_instead_of_leaving_the_hip_path(hip):
    """The convolution plan is the only eval-mode device path of the neck and head (round-4 review: a channel count the kernels do
    not take used to fall to PyTorch / MIOpen without a word)."""
    import logging

    from futuredet_amd import build_neck

    rpn = build_neck(dict(type="RPN", layer_nums=[1], ds_layer_strides=[1], ds_num_filters=[40], us_layer_strides=[1], us_num_filters=[40],
                          num_input_features=40, logger=logging.getLogger("RPN"))).cuda().eval()
    x = torch.randn((1, 40, 16, 16), device="cuda")
    with torch.no_grad():
        assert rpn.forward_modules(x).shape == (1, 40, 16, 16)  # (the nn.Module stack itself is fine)
        for dt in (torch.float32, torch.bfloat16):  # 40 channels: not a multiple of the fp32 (16) or bf16 (32) input granule
            rpn.compute_dtype = dt
            with pytest.raises(ValueError):
                rpn(x)
        rpn.compute_dtype = torch.float16
        with pytest.raises(ValueError):
            rpn(x)


# ------------------------------------------------------------------------------------------------ sweep assembly
def _write_sweep_files(tmp_path, g, case):
    rows = g[case + "_rows"]
    paths = []
    for s in range(len(rows) - 1):
        p = os.path.join(str(tmp_path), "%s_%d.bin" % (case, s))
        g[case + "_raw"][rows[s]:rows[s + 1]].tofile(p)
        paths.append(p)
    info = {"lidar_path": paths[0],
            "sweeps": [{"lidar_path": paths[s + 1], "transform_matrix": (g[case + "_mats"][s] if g[case + "_has"][s] else None),
                        "time_lag": float(g[case + "_lags"][s])} for s in range(len(paths) - 1)]}
    return info, len(paths)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_sweep_assembly_matches_reference_golden(hip, golden, tmp_path, case):
    """LoadPointCloudFromFile through the registry on .bin files, bit-exact vs the reference's own output."""
    from futuredet_amd import PIPELINES, build_from_cfg

    g = golden("sweeps.npz")
    info, nsweeps = _write_sweep_files(tmp_path, g, case)
    stage = build_from_cfg(dict(type="LoadPointCloudFromFile", dataset="NuScenesDataset"), PIPELINES)
    res, _ = stage({"lidar": {"nsweeps": nsweeps}, "painted": False}, info)
    out = res["lidar"]["combined"].cpu().numpy()
    ref = g[case + "_combined"]
    assert out.shape == ref.shape
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(res["lidar"]["points"].cpu().numpy(), g[case + "_points"])
    assert np.array_equal(res["lidar"]["times"].cpu().numpy(), g[case + "_times"])


def test_sweep_assembly_full_size_vs_oracle_and_padded_voxelization(hip):
    """10 sweeps x ~34k rows: bit-exact vs the oracle; the +inf padded output voxelizes like the trimmed one."""
    from oracle import ops as oops
    from futuredet_amd.synth import synthetic_cloud

    rng = np.random.default_rng(3)
    cloud = synthetic_cloud(seed=5, target_points=300000)
    S = 10
    chunks = np.array_split(cloud, S)
    raws, mats, lags = [], [], []
    for s, c in enumerate(chunks):
        raw = np.concatenate([c[:, :4], rng.integers(0, 32, (len(c), 1)).astype(np.float32)], axis=1)
        raw[: len(raw) // 50, :2] = rng.uniform(-1.2, 1.2, (len(raw) // 50, 2)).astype(np.float32)
        raws.append(np.ascontiguousarray(raw, np.float32))
        if s:
            a = rng.uniform(-0.05, 0.05)
            m = np.eye(4)
            m[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
            m[:3, 3] = rng.normal(0, [2.0, 0.5, 0.02])
            mats.append(None if s == 4 else m)
            lags.append(0.05 * s)
    ref = oops.assemble_sweeps(raws[0], raws[1:], mats, lags)
    order = [0] + [int(i) + 1 for i in np.random.default_rng(0).choice(S - 1, S - 1, replace=False)]
    rows = np.cumsum([0] + [len(raws[s]) for s in order])
    desc = hip.sweep_descriptors(rows, [None if s == 0 else mats[s - 1] for s in order], [0.0 if s == 0 else lags[s - 1] for s in order],
                                 [s != 0 for s in order])
    out, count = hip.assemble_sweeps(_dev(np.concatenate([raws[s] for s in order])), desc)
    n = int(count.item())
    assert n == len(ref) and n < out.shape[0]
    assert np.array_equal(out[:n].cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert bool(torch.isinf(out[n:]).all())
    vs, rg = [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0]
    a = hip.voxelize(out, vs, rg, 10, 160000, want_voxels=False, want_mean=True)
    b = hip.voxelize(out[:n].contiguous(), vs, rg, 10, 160000, want_voxels=False, want_mean=True)
    nv = int(a["num_voxels"].item())
    assert nv == int(b["num_voxels"].item()) and nv > 100000
    for k in ("coors", "num_points", "mean"):
        assert torch.equal(a[k][:nv], b[k][:nv])


def test_sweep_assembly_empty_and_single(hip):
    desc = hip.sweep_descriptors([0, 0], [None], [0.0], [False])
    out, count = hip.assemble_sweeps(torch.zeros((0, 5), device="cuda"), desc)
    assert out.shape == (0, 5) and int(count.item()) == 0
    raw = torch.tensor([[0.5, 0.5, 0.0, 1.0, 0.0], [2.0, 0.0, 0.0, 2.0, 0.0]], device="cuda")
    out, count = hip.assemble_sweeps(raw, hip.sweep_descriptors([0, 2], [None], [0.25], [True]))
    assert int(count.item()) == 1 and out[0].tolist() == [2.0, 0.0, 0.0, 2.0, 0.25]


# ------------------------------------------------------------------------------------------------ PointPillars
def _pillar_layers(g, name, n_layers):
    layers = []
    for i in range(n_layers):
        pre = "%s_sd_pfn_layers.%d." % (name, i)
        w, bw, bb, rm, rv = (g[pre + k] for k in ("linear.weight", "norm.weight", "norm.bias", "norm.running_mean", "norm.running_var"))
        scale = bw / np.sqrt(rv + 1e-3)
        layers.append((_dev(w), _dev(scale.astype(np.float32)), _dev((bb - rm * scale).astype(np.float32))))
    return layers


@pytest.mark.parametrize("name,n_layers,wd", [("two", 2, False), ("one", 1, True)])
def test_pillar_encode_and_scatter_match_reference_golden(hip, golden, name, n_layers, wd):
    """fd_pillar_encode / fd_pillar_scatter vs the reference's PillarFeatureNet + PointPillarsScatter outputs
    (fp32, 1e-3 of the feature scale), including pillars with all 20 slots full and the B=2 batch column."""
    g = golden("pillars.npz")
    geom = (0.2, 0.2, 0.2 / 2 + -6.4, 0.2 / 2 + -6.4)
    coors = _dev(g["coors"])
    M = len(g["num"])
    n_dev = torch.tensor([M - 7], dtype=torch.int32, device="cuda")  # the last 7 rows must stay untouched (zero)
    f = hip.pillar_encode(_dev(g["voxels"]), _dev(g["num"]), coors, n_dev, geom, _pillar_layers(g, name, n_layers), with_distance=wd)
    want = g[name + "_feats"]
    scale = np.abs(want).max()
    assert_close("pillar_encode %s vs reference golden" % name, f[:M - 7].cpu().numpy(), want[:M - 7], 1e-3)
    assert float(f[M - 7:].abs().max()) == 0.0
    f = hip.pillar_encode(_dev(g["voxels"]), _dev(g["num"]), coors, None, geom, _pillar_layers(g, name, n_layers), with_distance=wd)
    canvas = hip.pillar_scatter(f, coors, None, 2, 64, 64)
    assert tuple(canvas.shape) == (2, 64, 64, 64)
    assert np.abs(canvas.sum(1).cpu().numpy() - g[name + "_canvas_sum"]).max() <= 1e-3 * scale * 8
    assert_close("pillar_scatter %s channel 5 vs reference golden" % name, canvas[:, 5].cpu().numpy(), g[name + "_canvas_c5"], 1e-3)
    cl = hip.pillar_scatter(f, coors, None, 2, 64, 64, out_dtype=torch.bfloat16, channels_last=True)
    assert cl.is_contiguous(memory_format=torch.channels_last)
    assert float((cl.float() - canvas).abs().max()) <= 2 ** -8 * scale


def test_pp_rpn_matches_reference_golden(hip, golden):
    """RPN with the pp configs' deblock pattern (strided Conv2d, 1x1, transposed) through the module, fp32 (the bf16
    plan needs cin % 32 == 0 and is covered at full width in test_pp_dense_bf16_plans_vs_torch_modules)."""
    from futuredet_amd.necks import RPN

    g = golden("pillars.npz")
    rpn = RPN(layer_nums=[1, 2, 2], ds_layer_strides=[2, 2, 2], ds_num_filters=[16, 32, 64], us_layer_strides=[0.5, 1, 2],
              us_num_filters=[32, 32, 32], num_input_features=64)
    rpn.load_state_dict({k[7:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("rpn_sd_")})
    rpn = rpn.cuda().eval()
    want = g["rpn_out"]
    with torch.no_grad():
        y = rpn(_dev(g["rpn_in"]))
    assert_close("pp RPN vs reference golden", y.float().cpu().numpy(), want, 1e-3)


def test_pp_dense_bf16_plans_vs_torch_modules(hip):
    """Full-width pp RPN (strided / 1x1 / transposed deblocks) and the n3dtf head chain on the bf16 MFMA plan vs the
    fp32 torch modules with the same seeded weights."""
    from futuredet_amd import build_head, build_neck
    from futuredet_amd.configs import pointpillars_config
    from futuredet_amd.synth import seeded_state_dict

    cfg = pointpillars_config()
    neck = build_neck(cfg.model["neck"])
    neck.load_state_dict(seeded_state_dict(neck, 3), strict=False)
    head = build_head(cfg.model["bbox_head"])
    head.load_state_dict(seeded_state_dict(head, 4), strict=False)
    neck, head = neck.cuda().eval(), head.cuda().eval()
    x = torch.randn((1, 64, 64, 96), device="cuda", generator=torch.Generator("cuda").manual_seed(1)).relu_()
    with torch.no_grad():
        want = neck.forward_modules(x)
        neck.compute_dtype = torch.bfloat16
        got = neck(x)
        assert tuple(got.shape) == tuple(want.shape) == (1, 384, 16, 24)
        assert float((got.float() - want).abs().max()) <= 0.03 * float(want.abs().max())
        wp = head.forward_modules(want)
        head.compute_dtype = torch.bfloat16
        gp = head(want)
    assert len(gp) == len(wp) == 7
    for t in (0, 1, 6):
        for k in wp[t]:
            ref = wp[t][k]
            tol = 0.05 * max(1.0, float(ref.abs().max()))
            assert float((gp[t][k].float() - ref).abs().max()) <= tol, (t, k)


def test_pointpillars_end_to_end_vs_oracle(hip):
    """PointPillars (pp n3dtf config) on a 30k-point cloud: forward(example) and forward_points() vs the CPU oracle."""
    from futuredet_amd import build_detector
    from futuredet_amd.collate import collate_kitti_multi, example_to_device
    from futuredet_amd.configs import pointpillars_config
    from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims
    from futuredet_amd.voxelize import Voxelization
    from oracle import model as omodel

    cfg = pointpillars_config()
    net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = tame_box_dims(seeded_state_dict(net, 9))
    # raw coordinates (tens of metres) and intensities (0..255) enter the first Linear directly: scale it so the random
    # network stays in a sane range (otherwise every heat-map logit saturates and the scores are all exactly 1.0)
    sd["reader.pfn_layers.0.linear.weight"] = sd["reader.pfn_layers.0.linear.weight"] * 0.02
    net.load_state_dict(sd, strict=False)
    net = net.cuda().eval()
    onet = omodel.PointPillars(cfg.model["reader"], cfg.model["backbone"], cfg.model["neck"], cfg.model["bbox_head"],
                               test_cfg=cfg.test_cfg).eval()
    missing, unexpected = onet.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    clouds = [synthetic_cloud(seed=s, target_points=30000) for s in (2, 3)]
    vox = Voxelization(cfg=cfg.voxel_generator)
    examples = []
    for i, pts in enumerate(clouds):
        res, _ = vox({"mode": "val", "lidar": {"points": pts}}, None)
        v = res["lidar"]["voxels"]
        examples.append(dict(voxels=v["voxels"], coordinates=v["coordinates"], num_points=v["num_points"],
                             num_voxels=v["num_voxels"], shape=v["shape"], metadata={"token": i}))
    batch = collate_kitti_multi(examples)
    with torch.no_grad():
        want = onet(batch)
        dev_batch = example_to_device(batch, torch.device("cuda"))
        got = net(dev_batch, return_loss=False)
        fast = net.forward_points([_dev(c) for c in clouds], cfg.voxel_generator, padded=False)
    for b in range(2):
        w = _rows(want[b])
        assert len(w) > 50
        for tag, res in (("forward", got), ("forward_points", fast)):
            gt = _rows(res[b])
            # (the n3dtf chain of this config saturates scores to exactly 1.0 like the VoxelNet n3dtf case: count-bounded)
            gu, wu = gt[gt[:, 9] < 0.999], w[w[:, 9] < 0.999]
            bad = _match_detections(gu, wu)
            report("pointpillars e2e %s b%d detections" % (tag, b), float(bad), 0.02 * (len(gu) + len(wu)), "(unsaturated rows)")
            assert bad <= max(2, 0.02 * (len(gu) + len(wu))), (b, bad, len(gu), len(wu))


def test_spconv_ranges_partition_rows_and_balance_work(hip):
    """fd_spconv_ranges only steers scheduling: the table must be a monotone partition of the output rows (multiples of 32)
    whose ranges carry near-equal work on a rulebook with a strong density gradient, and the convolution must give the
    same bits with balanced ranges, equal-row ranges and one tile per workgroup."""
    rng = np.random.default_rng(5)
    K, n_out, cin, cout = 27, 128 * 173 + 50, 32, 32   # 2775 work blocks: every wave of the range kernel owns a segment, the last one ragged
    stride = (n_out + 63) // 64 * 64
    nbr = np.full((K, stride), -1, np.int32)
    dens = np.linspace(0.05, 0.9, n_out)  # sparse rows first, dense rows last
    m = rng.uniform(size=(K, n_out)) < dens[None, :]
    n_in = 5000
    nbr[:, :n_out] = np.where(m, rng.integers(0, n_in, (K, n_out)), -1)
    t = _dev(nbr)
    t.n_out = n_out
    ranges, n = hip.ranges_for(t, cin, cout)
    r = ranges.cpu().numpy()
    assert len(r) == n + 1 and r[0] == 0 and r[-1] == n_out and np.all(np.diff(r) >= 0) and np.all(r[:-1] % 8 == 0)
    # the table itself, restated: work of an 8-row block = 24 + 2 per pair + 1 per tap that has a pair in the block; range of a block =
    # floor(work midpoint * n / total) (one float64 multiply); a range starts at the first block that reaches it
    vb = nbr[:, :n_out] >= 0
    nblk = (n_out + 7) // 8
    padded = np.zeros((K, nblk * 8), bool)
    padded[:, :n_out] = vb
    per = padded.reshape(K, nblk, 8)
    wblk = (24 + 2 * per.sum((0, 2)) + per.any(2).sum(0)).astype(np.uint64)
    ex = np.concatenate([[0], np.cumsum(wblk)[:-1]]).astype(np.uint64)
    scale = np.float64(n) / (np.float64(2.0) * np.float64(wblk.sum()))
    rg = np.minimum((((2 * ex + wblk).astype(np.float64)) * scale).astype(np.int64), n - 1)
    want = np.full(n + 1, n_out, np.int64)
    prev = -1
    for bidx in range(nblk):
        for j in range(prev + 1, rg[bidx] + 1):
            want[j] = bidx * 8
        prev = int(rg[bidx])
    assert np.array_equal(r, want), "fd_spconv_ranges: the table differs from its definition"
    pairs = (nbr[:, :n_out] >= 0).sum(0)
    work = np.array([pairs[a:b].sum() for a, b in zip(r[:-1], r[1:])], np.float64)
    rows = np.diff(r)
    nz = work[rows > 0]
    assert nz.max() <= 1.25 * nz.mean() + 8 * 27, (nz.max(), nz.mean())  # equal work up to one 8-row block
    assert rows.max() > 2 * rows[rows > 0].min(), "sparse regions must get longer ranges than dense ones"
    x = torch.randn((n_in, cin), device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    w = torch.randn((K, cin, cout), generator=torch.Generator().manual_seed(2)) * 0.05
    wpk = hip.pack_spconv_weight(w).cuda()
    ys = [hip.spconv_apply(x, wpk, None, t, n_out, cout, balanced=True)]
    ys.append(hip.spconv_apply(x, wpk, None, t, n_out, cout, balanced=False))
    try:
        hip.set_tuning("v2_ranges_per_cu", 1)   # few long ranges: several chunks per workgroup (prefetch path)
        ys.append(hip.spconv_apply(x, wpk, None, t, n_out, cout, balanced=False))
        t2 = _dev(nbr)
        t2.n_out = n_out
        ys.append(hip.spconv_apply(x, wpk, None, t2, n_out, cout, balanced=True))
    finally:
        hip.set_tuning("v2_ranges_per_cu", 0)
    for y in ys[1:]:
        assert torch.equal(ys[0], y), "the work distribution must not change a single bit"
    ref = torch.zeros((n_out, cout))
    xc, nb = x.cpu(), torch.from_numpy(nbr[:, :n_out].astype(np.int64))
    for k in range(K):
        v = nb[k] >= 0
        ref[v] += xc[nb[k][v]] @ w[k]
    assert_close("spconv with work-balanced ranges vs dense torch reference", ys[0].cpu().numpy(), ref.numpy(), 1e-4)


# ------------------------------------------------------------------------------------------------ forecast association
class _Box(object):
    def __init__(self, c, v, tag):
        self.center, self.velocity, self.tag = np.array(c, np.float64), np.array(v, np.float64), tag


@pytest.mark.parametrize("case,cls", [("car", "car"), ("ped", "pedestrian"), ("sparse", "car"), ("empty", "car")])
def test_forecast_tracker_matches_reference_golden(hip, golden, case, cls):
    """futuredet_amd.forecast.tracker / match_boxes (fd_forecast_chains) vs the reference's own functions: identical
    trajectory membership and order, identical float64 centres."""
    from futuredet_amd import forecast

    g = golden("forecast.npz")
    T = 7
    ret_boxes = [[_Box(c, v, (t, j)) for j, (c, v) in enumerate(zip(g["%s_centers_%d" % (case, t)], g["%s_velocity_%d" % (case, t)]))]
                 for t in range(T)]
    traj = forecast.tracker(cls, list(g[case + "_time"]), ret_boxes)
    tags = np.asarray([[b.tag[1] for b in tr] for tr in traj], np.int64).reshape(-1, T)
    cents = np.asarray([[b.center for b in tr] for tr in traj], np.float64).reshape(-1, T, 3)
    assert np.array_equal(tags, g[case + "_traj_tags"])
    assert np.array_equal(cents, g[case + "_traj_centers"])
    if case + "_match_tags" in g:
        mb = forecast.match_boxes(ret_boxes)
        assert np.array_equal(np.asarray([[b.tag[1] for b in row] for row in mb], np.int64), g[case + "_match_tags"])
    assert forecast.tracker("truck", list(g[case + "_time"]), ret_boxes) == []


def test_det_to_global_boxes_match_reference_golden(hip, golden):
    """fd_det_to_global_boxes vs the reference's _second_det_to_nusc_box + _lidar_nusc_box_to_global (forecast2.npz).  float64
    on the device (fma chains, device sin/cos) vs numpy on the host: |d| <= 1e-9 * max(1, |ref|); the float32 yaw flip and
    the velocity triple are exact."""
    from futuredet_amd import forecast

    g = golden("forecast2.npz")
    det = {"box3d_lidar": torch.from_numpy(g["box3d"]), "scores": torch.from_numpy(g["scores"]), "label_preds": torch.from_numpy(g["labels"])}
    c, q, v, s = forecast.det_arrays(det)
    assert np.array_equal(c, g["lidar_center"]) and np.array_equal(v, g["lidar_velocity"]) and np.array_equal(s, g["lidar_size"])
    assert_close("det -> lidar boxes: quaternion", q, g["lidar_quat"], 1e-12)
    cs, pose = (g["cs_rotation"], g["cs_translation"]), (g["pose_rotation"], g["pose_translation"])
    c, q, v, s = forecast.det_arrays(det, cs, pose)
    assert_close("det -> global boxes: center", c, g["global_center"], 1e-9)
    assert_close("det -> global boxes: quaternion", q, g["global_quat"], 1e-9)
    assert_close("det -> global boxes: velocity", v, g["global_velocity"], 1e-9)
    boxes = forecast._second_det_to_nusc_box(det)
    assert len(boxes) == len(g["box3d"]) and boxes[3].center.dtype == np.float32 and float(boxes[3].score) == float(g["scores"][3])
    assert forecast._second_det_to_nusc_box({k: t[:0] for k, t in det.items()}) == []


@pytest.mark.parametrize("mode", ["velocity_constant", "velocity_forward", "velocity_reverse", "velocity_dense"])
def test_forecast_boxes_match_reference_golden(hip, golden, mode):
    """futuredet_amd.forecast.forecast_boxes (per-step split, global transform, match_boxes / tracker, constant-velocity
    roll-out) vs the reference's forecast_boxes run on the same detections: same trajectories in the same order, labels and
    scores identical, float64 centres / orientations / velocities within 1e-9."""
    from futuredet_amd import forecast

    g = golden("forecast2.npz")
    det = {"box3d_lidar": torch.from_numpy(g["box3d"]).cuda(), "scores": torch.from_numpy(g["scores"]).cuda(),
           "label_preds": torch.from_numpy(g["labels"]).cuda()}
    cs, pose = (g["cs_rotation"], g["cs_translation"]), (g["pose_rotation"], g["pose_translation"])
    ret = forecast.forecast_boxes(det, list(g["time"]), cs, pose, 7, mode, "car")
    want_c = g[mode + "_center"]
    assert len(ret) == len(want_c) and all(len(tr) == 7 for tr in ret)
    assert np.array_equal(np.array([[b.label for b in tr] for tr in ret], np.int64), g[mode + "_label"])
    assert np.array_equal(np.array([[float(b.score) for b in tr] for tr in ret]), g[mode + "_score"])
    assert_close("forecast_boxes %s centres" % mode, np.array([[b.center for b in tr] for tr in ret], np.float64), want_c, 1e-9)
    assert_close("forecast_boxes %s orientations" % mode, np.array([[b.orientation.elements for b in tr] for tr in ret]), g[mode + "_quat"], 1e-9)
    assert_close("forecast_boxes %s velocities" % mode, np.array([[b.velocity for b in tr] for tr in ret]), g[mode + "_velocity"], 1e-9)
    assert forecast.forecast_boxes(det, [0.5, 0.0, 0.5, 0.5, 0.5, 0.5], cs, pose, 7, mode, "car") == []  # a stale step (:402-403,419-420)


def test_forecast_postprocess_and_sparse_modes_match_reference_golden(hip, golden):
    """The rest of forecast_boxes (VERDICT r2 missing #3): velocity_dense with postprocess=True -- process_trajectories
    (nuscenes.py:341-382), the nearest-library-trajectory search on the device (fd_nearest_rows) -- gives the reference's centres
    (1e-9), and the three velocity_sparse_* modes end the way they end in the reference (the fixture records the reference's own
    exception: they are dead code there, nuscenes.py:422-429,468-469)."""
    from futuredet_amd import forecast

    g = golden("forecast2.npz")
    det = {"box3d_lidar": torch.from_numpy(g["box3d"]).cuda(), "scores": torch.from_numpy(g["scores"]).cuda(),
           "label_preds": torch.from_numpy(g["labels"]).cuda()}
    cs, pose = (g["cs_rotation"], g["cs_translation"]), (g["pose_rotation"], g["pose_translation"])
    ret = forecast.forecast_boxes(det, list(g["time"]), cs, pose, 7, "velocity_dense", "car", train_dist=g["pp_train_dist"], postprocess=True)
    assert len(ret) == len(g["pp_center"])
    assert_close("forecast_boxes velocity_dense + postprocess centres", np.array([[b.center for b in tr] for tr in ret], np.float64), g["pp_center"], 1e-9)
    # the search itself against numpy on a larger library, duplicates included (first minimum wins)
    rng = np.random.default_rng(5)
    lib = rng.normal(0, 3, (5000, 24))
    lib[1234] = lib[77]
    q = np.concatenate([lib[[77, 4999, 0]] + 1e-3, rng.normal(0, 3, (200, 24))])
    got = hip.nearest_rows(torch.from_numpy(lib).cuda(), torch.from_numpy(q).cuda()).cpu().numpy()
    want = np.argmin(((lib[None] - q[:, None]) ** 2).sum(-1), axis=1)
    assert np.array_equal(got, want) and got[0] == 77
    assert list(g["sparse_mode_exception"]) == ["TypeError"] * 3
    for mode in ("velocity_sparse_forward", "velocity_sparse_reverse", "velocity_sparse_match"):
        with pytest.raises(TypeError):
            forecast.forecast_boxes(det, list(g["time"]), cs, pose, 7, mode, "car")
    with pytest.raises(AssertionError):
        forecast.forecast_boxes(det, list(g["time"]), cs, pose, 7, "no_such_mode", "car")


def test_multi_future_matches_reference_golden(hip, golden):
    """forecast ids (connected components of the < 0.25 m graph, numbered like networkx enumerates them) and the copied
    scores vs the reference's multi_future; a 0.2 m chain must end up in one component, other classes are dropped."""
    from futuredet_amd import forecast

    g = golden("forecast2.npz")
    fb = {"tokA": [{"sample_token": "tokA", "translation": c.tolist(), "detection_name": "car" if car else "pedestrian",
                    "detection_score": float(ds), "forecast_score": float(fs), "forecast_id": -1,
                    "forecast_boxes": [{"detection_score": 0.0, "forecast_score": 0.0, "forecast_id": -1} for _ in range(3)]}
                   for c, car, ds, fs in zip(g["mf_translation"], g["mf_is_car"], g["mf_det_score"], g["mf_fc_score"])], "tokB": []}
    res = forecast.multi_future(fb, "car")
    assert np.array_equal(np.array([b["forecast_id"] for b in res["tokA"]], np.int64), g["mf_ids"])
    assert np.array_equal(np.array([[s["forecast_id"] for s in b["forecast_boxes"]] for b in res["tokA"]], np.int64), g["mf_sub_ids"])
    assert np.array_equal(np.array([[s["detection_score"] for s in b["forecast_boxes"]] for b in res["tokA"]]), g["mf_sub_det"])
    assert res["tokB"] == [] and len(forecast.forecast_ids(np.zeros((0, 3)))) == 0


# ------------------------------------------------------------------------------------------------ edge cases
def test_forward_points_empty_and_ragged_batch(hip):
    """An empty cloud, a cloud entirely outside the range and a normal cloud in one batch: no crash, finite output for
    the empty samples, and the normal sample's detections match its single-sample run (samples are independent)."""
    from futuredet_amd import build_detector
    from futuredet_amd.configs import centerpoint_config
    from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims

    cfg = centerpoint_config("forecast_n0")
    net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    net.load_state_dict(tame_box_dims(seeded_state_dict(net, 7)), strict=False)
    net = net.cuda().eval()
    good = _dev(synthetic_cloud(seed=4, target_points=20000))
    empty = torch.zeros((0, 5), device="cuda")
    far = torch.full((100, 5), 500.0, device="cuda")
    solo = net.forward_points([good], cfg.voxel_generator, padded=False)[0]
    batch = net.forward_points([empty, good, far], cfg.voxel_generator, padded=False)
    assert len(batch) == 3
    assert len(solo["scores"]) > 0
    got = torch.cat([batch[1]["box3d_lidar"], batch[1]["scores"][:, None], batch[1]["label_preds"][:, None].float()], 1).cpu().numpy()
    want = torch.cat([solo["box3d_lidar"], solo["scores"][:, None], solo["label_preds"][:, None].float()], 1).cpu().numpy()
    _attribute("ragged batch: sample 1 of 3 vs its single-sample run", got, want, cfg.test_cfg)  # (the dense convs may tile differently at B=3)
    for b in (0, 2):  # an all-zero BEV map still decodes whatever the biases alone produce; it must be finite and bounded
        assert batch[b]["box3d_lidar"].shape[1] == 9 and bool(torch.isfinite(batch[b]["box3d_lidar"]).all())
    none = net.forward_points([empty], cfg.voxel_generator, padded=False)
    assert len(none) == 1 and bool(torch.isfinite(none[0]["scores"]).all())


# ------------------------------------------------------------------------------------------------ C ABI error behaviour
def test_c_abi_reports_errors_instead_of_exiting(hip):
    """The reference extension calls exit(-1) on bad input (iou3d_nms.cpp:14-38); this library returns a negative status
    and a message, for every family of entry points, and stays usable afterwards."""
    import ctypes

    from futuredet_amd import lib

    L = lib.load()
    x = torch.zeros((64, 16), device="cuda")
    cases = [
        ("fd_voxelize", lambda: L.fd_voxelize(x.data_ptr(), 64, None, 99, None, None, 10, 100, 0, None, None, 0, None, 3, None, None, None, 0, None)),
        ("fd_spconv_apply", lambda: L.fd_spconv_apply(x.data_ptr(), 64, x.data_ptr(), None, None, 0, x.data_ptr(), 64, None, 0, 99, 64, None, 0,
                                                     16, 16, 0, x.data_ptr(), None)),
        ("fd_rotated_nms", lambda: L.fd_rotated_nms(None, 10, ctypes.c_float(0.2), None, None, None, 0, None)),
        ("fd_conv2d_nhwc_bf16", lambda: L.fd_conv2d_nhwc_bf16(x.data_ptr(), 1, 8, 8, 7, x.data_ptr(), None, 16, 3, 1, 1, 1, x.data_ptr(), 16, 0, 1, 1,
                                                             0, 0, None)),
        ("fd_conv2d_nhwc_f32", lambda: L.fd_conv2d_nhwc_f32(x.data_ptr(), 1, 8, 8, 16, x.data_ptr(), None, 16, 3, 1, 1, 1, x.data_ptr(), 16, 0, 1, 1,
                                                            0, 0, 99, None)),
        ("fd_sweep_assemble", lambda: L.fd_sweep_assemble(x.data_ptr(), 5, 9, 10, x.data_ptr(), 1, ctypes.c_float(1.0), x.data_ptr(), x.data_ptr(),
                                                         None, 0, None)),
        ("fd_pillar_encode", lambda: L.fd_pillar_encode(x.data_ptr(), x.data_ptr(), x.data_ptr(), None, 4, 99, 5, 0, ctypes.c_float(1), ctypes.c_float(1),
                                                       ctypes.c_float(0), ctypes.c_float(0), x.data_ptr(), x.data_ptr(), x.data_ptr(), 64, None, None,
                                                       None, 0, 0, x.data_ptr(), 64, None)),
        ("fd_forecast_groups", lambda: L.fd_forecast_groups(x.data_ptr(), 9000, ctypes.c_double(0.25), x.data_ptr(), None)),
        ("fd_det_to_global_boxes", lambda: L.fd_det_to_global_boxes(x.data_ptr(), 4, (ctypes.c_double * 4)(1, 0, 0, 0), None, None, None, x.data_ptr(),
                                                                   x.data_ptr(), x.data_ptr(), x.data_ptr(), None)),
        ("fd_forecast_chains", lambda: L.fd_forecast_chains(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 99, 8, ctypes.c_double(1.0),
                                                           x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                                           x.data_ptr(), None)),
    ]
    for name, call in cases:
        status = call()
        assert status < 0, name
        msg = L.fd_last_error()
        assert msg and name.encode() in msg, (name, msg)
    # the decode's NMS configuration (ABI 8): an unknown nms_kind, a radius table that does not divide the groups, a negative radius
    maps = [torch.zeros((2, c, 16, 16), device="cuda") for c in (1, 2, 1, 3, 2)]
    for kw in (dict(nms_kind=2), dict(nms_kind=1, n_radius=0), dict(nms_kind=1, n_radius=3), dict(nms_kind=1, n_radius=2, radius0=-1.0)):
        cfg = hip.make_decode_cfg(16, 16, TEST_CFG)
        cfg.nms_kind, cfg.n_radius = kw.get("nms_kind", 0), kw.get("n_radius", 0)
        cfg.circle_radius[0] = kw.get("radius0", 1.0)
        with pytest.raises(lib.FutureDetHipError, match="fd_centerpoint_decode"):
            hip.centerpoint_decode(*maps, cfg)
    # workspace too small is its own code, and the library still works afterwards
    pts = torch.rand((100, 5), device="cuda")
    out = hip.voxelize(pts, [0.5, 0.5, 0.5], [0, 0, 0, 1, 1, 1], 4, 64)
    assert 0 < int(out["num_voxels"].item()) <= 8
    with pytest.raises(lib.FutureDetHipError):
        hip.voxelize(torch.rand((10, 5)), [0.5, 0.5, 0.5], [0, 0, 0, 1, 1, 1], 4, 64)  # CPU tensor: no CPU path


def _build_pair(variant, class_name="car", seed=7, **cfg_kw):
    """(cfg, HIP detector on the GPU, CPU oracle detector) with the same seeded weights."""
    from futuredet_amd import build_detector
    from futuredet_amd.configs import centerpoint_config
    from futuredet_amd.synth import seeded_state_dict, tame_box_dims
    from oracle import model as omodel

    cfg = centerpoint_config(variant, class_name, **cfg_kw)
    net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = tame_box_dims(seeded_state_dict(net, seed))  # box sizes of metres, not exp(150) m (see synth.tame_box_dims)
    net.load_state_dict(sd, strict=False)
    net = net.cuda().eval()
    onet = omodel.VoxelNet(cfg.model["reader"], cfg.model["backbone"], cfg.model["neck"], cfg.model["bbox_head"],
                           test_cfg=cfg.test_cfg).eval()
    missing, unexpected = onet.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    return cfg, net, onet


def _grid_of(vg):
    return np.round((np.array(vg["range"][3:], np.float32) - np.array(vg["range"][:3], np.float32)) / np.array(vg["voxel_size"], np.float32)).astype(np.int64)


def _oracle_run(cfg, onet, cloud):
    """The oracle on one cloud: (voxels, coords(z,y,x), num_points, backbone BEV, neck BEV, detections dict)."""
    from oracle import ops as oops

    vg = cfg.voxel_generator
    v, c, n = oops.points_to_voxel(cloud, vg["voxel_size"], vg["range"], vg["max_points_in_voxel"], True, vg["max_voxel_num"][1])
    ex = dict(voxels=torch.from_numpy(v), coordinates=torch.from_numpy(np.pad(c, ((0, 0), (1, 0)))), num_points=torch.from_numpy(n),
              num_voxels=torch.tensor([len(n)]), shape=np.array([_grid_of(vg)]), metadata=[None])
    with torch.no_grad():
        feats = onet.reader(ex["voxels"], ex["num_points"])
        bb, _ = onet.backbone(feats, ex["coordinates"], 1, ex["shape"][0])
        bev = onet.neck(bb)
        preds = onet.bbox_head(bev, None)
        det = onet.bbox_head.predict(ex, preds, cfg.test_cfg)[0]
    # the nms_pre_max-th best candidate score (standard head: one heat-map shared by the steps), None when fewer cells qualify
    sc = torch.sigmoid(preds[0]["hm"].float()).flatten()
    k = int(cfg.test_cfg["nms"]["nms_pre_max_size"])
    det["topk_cut"] = float(torch.topk(sc, k).values[-1]) if int((sc > cfg.test_cfg["score_threshold"]).sum()) > k else None
    return v, c, n, bb, bev, det


def _hip_maps(net, cfg, v, c, n):
    """Backbone and neck BEV maps of the HIP modules on given voxels (the device voxelizer is checked bit-exact elsewhere)."""
    with torch.no_grad():
        feats = net.reader(_dev(v).float(), _dev(n))
        bb, _ = net.backbone(feats, _dev(np.pad(c, ((0, 0), (1, 0)))), 1, [int(g) for g in _grid_of(cfg.voxel_generator)])
        return bb, net.neck(bb)


BF16_ULP = 8e-3  # one bf16 ulp relative to the value (2^-8 .. 2^-7): the bound for a layer fed IDENTICAL inputs


def _bf16_example(cfg, cloud):
    from oracle import ops as oops

    vg = cfg.voxel_generator
    v, c, n = oops.points_to_voxel(cloud, vg["voxel_size"], vg["range"], vg["max_points_in_voxel"], True, vg["max_voxel_num"][1])
    ex = dict(voxels=torch.from_numpy(v), coordinates=torch.from_numpy(np.pad(c, ((0, 0), (1, 0)))), num_points=torch.from_numpy(n),
              num_voxels=torch.tensor([len(n)]), shape=np.array([_grid_of(vg)]), metadata=[None])
    return v, c, n, ex


# Achieved end-to-end deviation of the bf16 pipeline from the bf16 oracle, round 3 on MI355X (profiles/round3_parity_report.txt):
# (mean, fraction of active elements beyond 2e-2) per map.  The gates are 2 x these per configuration (the flat
# 3e-2 / 30 % of round 3 would not have noticed a 2 x regression), with floors for the maps whose achieved figure is ~0.
_BF16_ACHIEVED = {
    "30k n3": dict(backbone=(1.8e-3, 1.14e-2), neck=(2.94e-3, 1.68e-2), reg=(4.65e-3, 3.15e-2), height=(1.97e-3, 1.1e-3), dim=(6e-5, 0.0), rot=(3.7e-3, 2.35e-2),
                   vel=(3.67e-3, 1.87e-2), hm=(2.77e-3, 2.08e-2)),
    "full size config 3": dict(backbone=(4.17e-3, 3.24e-2), neck=(7.77e-3, 6.62e-2), reg=(1.224e-2, 0.123), height=(3.55e-3, 4.2e-3), dim=(3.2e-4, 0.0),
                               rot=(9.52e-3, 8.58e-2), vel=(8.74e-3, 7.4e-2), hm=(9.09e-3, 8.84e-2)),
    "config 4 sample 3": dict(backbone=(4.18e-3, 3.24e-2), neck=(7.91e-3, 6.87e-2), reg=(1.244e-2, 0.127), height=(3.55e-3, 4.9e-3), dim=(3.4e-4, 0.0),
                              rot=(9.89e-3, 9.33e-2), vel=(8.87e-3, 7.64e-2), hm=(9.12e-3, 9.03e-2)),
    "full size config 5": dict(backbone=(4.25e-3, 3.39e-2), neck=(6.99e-3, 5.84e-2), reg=(1.082e-2, 0.107), height=(3.34e-3, 4.9e-3), dim=(2.7e-4, 0.0),
                               rot=(8.73e-3, 7.88e-2), vel=(7.96e-3, 6.54e-2), hm=(8.17e-3, 8.05e-2)),
    "street scene": dict(backbone=(6.28e-3, 6.78e-2), neck=(7.64e-3, 8.86e-2), reg=(1.205e-2, 0.155), height=(3.39e-3, 7.1e-3), dim=(2.7e-4, 0.0),
                         rot=(1.054e-2, 0.132), vel=(9.33e-3, 0.108), hm=(1.46e-3, 9.4e-3)),
}


def _bf16_gate(tag, kind):
    """(mean, fraction) gates = 2 x the achieved figures of this configuration, floors 1e-3 / 5e-3; unknown tags: the worst
    configuration's figures."""
    for key, d in _BF16_ACHIEVED.items():
        if tag.startswith(key):
            m, f = d[kind]
            break
    else:
        m = max(d[kind][0] for d in _BF16_ACHIEVED.values())
        f = max(d[kind][1] for d in _BF16_ACHIEVED.values())
    return max(2.0 * m, 1e-3), max(2.0 * f, 5e-3)


def _check_bf16(tag, cfg, net, onet, cloud, got, min_rows=20):
    """The HIP bf16 path end to end against the oracle's bf16 configuration (oracle/bf16.py: weights and per-layer activations
    rounded to bf16, fp32 accumulate) on one cloud.  Element-wise equality to a tolerance cannot be asked of ~45 chained bf16
    layers: the two implementations sum in different orders, so a pre-rounding value near a rounding boundary rounds differently
    (one ulp, 2^-8 relative), and downstream layers amplify such flips (measured: 1 ulp after stage 1, 5e-2 after stage 2,
    0.3 on a few elements of the 128-channel stage, 0.8 on single elements of the neck output).  What IS asserted here: the
    DISTRIBUTION of the deviation (mean and the fraction of active elements beyond 2e-2 * max(1, |ref|), each gated at 2 x the
    figure this configuration achieved in round 3, _BF16_ACHIEVED: mean 2e-3 .. 1.2e-2, fraction 1 .. 15 %, growing with the size
    of the cloud) for the backbone BEV, the neck output and every head map, and that the detections
    agree except for attributed near-threshold decisions.  The element-wise bound (one ulp) is asserted per LAYER with identical
    inputs in test_bf16_every_layer_teacher_forced; the decode on identical maps in test_bf16_decode_on_identical_maps.
    The distance to the fp32 oracle is reported (it measures bf16 itself)."""
    from oracle import bf16 as obf
    from oracle import ops as oops

    v, c, n, ex = _bf16_example(cfg, cloud)
    obb, obev, opreds, odet = obf.run(onet, ex, cfg.test_cfg)
    with torch.no_grad():
        feats32 = onet.reader(ex["voxels"], ex["num_points"])
        fbb, _ = onet.backbone(feats32, ex["coordinates"], 1, ex["shape"][0])
        fbev = onet.neck(fbb)
    bb, x = _hip_maps(net, cfg, v, c, n)
    with torch.no_grad():
        preds = net.bbox_head(x)

    def dist(name, kind, g, r):
        g, r = np.asarray(g, np.float64), np.asarray(r, np.float64)
        d = np.abs(g - r) / np.maximum(1.0, np.abs(r))
        act = (g != 0) | (r != 0)  # (the BEV map is mostly empty cells, identical zeros on both sides)
        d = d[act] if act.any() else d.reshape(-1)
        mean, frac, mx = float(d.mean()), float((d > 2e-2).mean()), float(d.max())
        gm, gf = _bf16_gate(tag, kind)
        report(tag + " bf16 %s: mean" % name, mean, gm, "(frac > 2e-2: %.2e of gate %.2e, max %.3f, active elements %d)" % (frac, gf, mx, d.size))
        assert mean <= gm and frac <= gf, (name, mean, gm, frac, gf, mx)

    dist("backbone BEV vs bf16 oracle", "backbone", bb.float().cpu().numpy(), obb.numpy())
    dist("neck output vs bf16 oracle", "neck", x.float().cpu().numpy(), obev.numpy())
    for ti, (pg, po) in enumerate(zip(preds, opreds)):
        for k in po:
            if k != "feats":
                dist("head task %d %s vs bf16 oracle" % (ti, k), k, pg[k].float().cpu().numpy(), po[k].numpy())
    scale = float(fbev.abs().max())
    report(tag + " bf16 neck output vs FP32 oracle (max, of scale; reported, not gated)", float((x.float().cpu() - fbev).abs().max()) / scale, 1.0)
    return _attribute_bf16_detections(tag, cfg, _rows(got), _rows(odet[0]), opreds, preds, min_rows)


def _attribute_bf16_detections(tag, cfg, g, w, opreds, dpreds, min_rows=20):
    """End-to-end bf16 detections: every detection of the bf16 oracle must be found by the device (same time step, centre within
    0.3 m) or be ATTRIBUTED to the map deviation at its own cell: the device's score there fell below the score threshold / the
    nms_pre_max cut, a device detection overlapping it above the NMS threshold took its place, or it fell out of the
    nms_post_max best.  (Row-by-row equality is not available: yaw = atan2 of two maps that deviate by up to 0.25 and velocities
    deviate by up to 0.5 on single cells.)  Returns the counts per category; asserts that nothing stays unexplained."""
    from oracle import ops as oops

    tc = cfg.test_cfg
    thr, iou_thr = tc["score_threshold"], tc["nms"]["nms_iou_threshold"]
    pre_max, post_max = int(tc["nms"]["nms_pre_max_size"]), int(tc["nms"]["nms_post_max_size"])
    assert len(w) >= min_rows
    dense = len(opreds) > 1
    cat = dict(found=0, score=0, pre_cut=0, range=0, nms=0, post_cut=0, unexplained=0)
    cache = {}
    margins = []  # how far above the oracle's own pre-max cut the detections lost to the device's cut were
    lost_cells = set()  # (task, cell) of the detections not found: the standard head shares ONE heat-map and one box per cell between its
                        # 7 time steps, so a single cell that falls out of the device's top-1000 costs 7 detections -- which is why the
                        # saturated configurations lose multiples of 7 (round 3: 21 = 3 cells x 7 steps on configs 3 and 5)
    for row in w:
        lab = int(row[10])
        cand = g[g[:, 10] == lab]
        if len(cand) and np.abs(cand[:, :2] - row[:2]).max(1).min() < 0.3:
            cat["found"] += 1
            continue
        ti = lab if dense else 0
        if ti not in cache:  # per task: both pipelines' decoded centres, score maps and candidate masks, and both pre-max cuts
            rng_lim = [float(v) for v in tc["post_center_limit_range"]]

            def decode(pd):
                """(cx, cy, score, candidate mask) of every cell as predict sees it: the candidates of the top-k are the cells with
                score > threshold AND centre inside post_center_limit_range (center_head.py:709-716 masks before
                box_torch_ops.py:259-261 sorts and cuts) -- round 4 computed the cuts over all cells and its margins came out negative."""
                reg = pd["reg"][0].float().cpu().permute(1, 2, 0)
                H, W = reg.shape[:2]
                ys, xs = torch.meshgrid([torch.arange(0, H), torch.arange(0, W)], indexing="ij")
                cx = (xs + reg[..., 0]) * tc["out_size_factor"] * tc["voxel_size"][0] + tc["pc_range"][0]
                cy = (ys + reg[..., 1]) * tc["out_size_factor"] * tc["voxel_size"][1] + tc["pc_range"][1]
                cz = pd["height"][0, 0].float().cpu()
                sc = torch.sigmoid(pd["hm"][0].float().cpu()).max(0).values
                ok = (sc > thr) & (cx >= rng_lim[0]) & (cy >= rng_lim[1]) & (cz >= rng_lim[2]) & (cx <= rng_lim[3]) & (cy <= rng_lim[4]) & (cz <= rng_lim[5])
                cut = float(torch.topk(sc[ok], pre_max).values[-1]) if int(ok.sum()) > pre_max else None
                return cx.numpy().ravel(), cy.numpy().ravel(), sc.numpy().ravel(), ok.numpy().ravel(), cut

            cache[ti] = decode(opreds[ti]) + decode(dpreds[ti])
        cx, cy, so, ok_o, ko, cxd, cyd, sd, ok_d, kth = cache[ti]
        cell = int(np.argmin(np.abs(cx - row[0]) + np.abs(cy - row[1]) + 10.0 * np.abs(so - row[9])))
        assert abs(cx[cell] - row[0]) < 1e-3 and abs(cy[cell] - row[1]) < 1e-3, "oracle detection not found among its own cells"
        assert ok_o[cell] and (ko is None or so[cell] >= ko), "an oracle detection must be one of the oracle's own pre-max candidates"
        lost_cells.add((ti, cell))
        s_dev, s_ora = float(sd[cell]), float(so[cell])
        # A score / pre-max decision may differ only by as much as the two pipelines' scores differ AT THIS CELL (measured, not a flat
        # slack): the oracle had the cell above the bar by (s_ora - bar); the device has it below iff its score moved down by more.
        dev_move = abs(s_dev - s_ora) + 1e-6
        in_dev_topk = bool(ok_d[cell]) and (kth is None or s_dev > kth)  # the cell IS one of the device's pre-max candidates
        if s_dev <= thr and s_ora - thr <= dev_move:
            cat["score"] += 1
        elif not in_dev_topk and kth is not None and s_dev <= kth and (ko is None or (s_ora - ko) <= dev_move + abs(kth - ko)):
            cat["pre_cut"] += 1
            margins.append((s_ora - ko) if ko is not None else 0.0)
        elif not ok_d[cell] and s_dev > thr and max(abs(float(cxd[cell]) - float(cx[cell])), abs(float(cyd[cell]) - float(cy[cell]))) + 1e-3 >= min(
                abs(float(cx[cell]) - rng_lim[0]), abs(float(cx[cell]) - rng_lim[3]), abs(float(cy[cell]) - rng_lim[1]), abs(float(cy[cell]) - rng_lim[4])):
            cat["range"] += 1  # the device's centre of this cell left post_center_limit_range by no more than the two centres differ
        elif len(cand) and float(oops.boxes_iou_bev(nms_layout(row[None, :9]), nms_layout(cand[:, :9])).max()) > iou_thr - 6e-2:
            cat["nms"] += 1
        elif len(cand) >= post_max and float(cand[:, 9].min()) >= s_dev - 1.5e-2:
            cat["post_cut"] += 1
        else:
            cat["unexplained"] += 1
    report(tag + " bf16 detections vs bf16 oracle: not found", float(len(w) - cat["found"]), float(len(w)),
           "(of %d: score %d, pre-max cut %d, range %d, NMS %d, post-max cut %d, unexplained %d; %d distinct cells; pre-max losses sat %.1e .. %.1e above the oracle's cut)"
           % (len(w), cat["score"], cat["pre_cut"], cat["range"], cat["nms"], cat["post_cut"], cat["unexplained"], len(lost_cells), min(margins) if margins else 0.0,
              max(margins) if margins else 0.0))
    assert cat["unexplained"] == 0, cat
    assert all(m >= 0.0 for m in margins), ("a detection lost to the pre-max cut must sit AT or ABOVE the oracle's own cut", margins)
    # (a scene with a handful of detections: ONE attributed cell is 7 rows of 21 -- the share rule applies beyond one cell)
    assert cat["found"] >= 0.8 * len(w) or len(lost_cells) <= 1, (cat, len(lost_cells))
    return cat


def _rows_by_coord(hip_coords, ora_coords):
    """permutation p with hip row r  <->  oracle row p[r] (both [n,4] b,z,y,x)"""
    key = lambda q: ((q[:, 0].astype(np.int64) * 64 + q[:, 1]) * 8192 + q[:, 2]) * 8192 + q[:, 3]  # noqa: E731
    kh, ko = key(hip_coords), key(ora_coords)
    so = np.argsort(ko)
    pos = np.searchsorted(ko[so], kh)
    assert np.array_equal(ko[so][pos], kh), "active sets differ"
    return so[pos]


@pytest.mark.parametrize("variant,points", [("forecast_n3", 300000), ("forecast_n3dtf", 40000)])
def test_bf16_every_layer_teacher_forced(hip, variant, points):
    """bf16 parity with teeth: every fused layer of the bf16 configuration -- the 21 sparse convolutions (with their
    folded BatchNorm, residual and ReLU), every RPN convolution / deblock, the head's shared, forecast, first and final convolutions --
    is run on the DEVICE with the INPUT the bf16 oracle had for that layer (oracle/bf16.py trace), and its output must equal the
    oracle's output element-wise within ONE bf16 ulp (8e-3 * max(1, |ref|)): with identical inputs only the fp32 summation order
    differs, which can move the final rounding by an ulp and no further.  Full-size cloud (config 3) and the forecast_feature head."""
    from futuredet_amd import hip_ops, sparse as fsparse
    from futuredet_amd.dense_bf16 import HeadPlan, RPNPlan
    from futuredet_amd.synth import synthetic_cloud
    from oracle import bf16 as obf

    cfg, net, onet = _build_pair(variant)
    net.set_precision(torch.bfloat16)
    cloud = synthetic_cloud(seed=0, target_points=points)
    v, c, n, ex = _bf16_example(cfg, cloud)
    with obf.tracing() as tr:
        obf.run(onet, ex, cfg.test_cfg)
    sp = [r for r in tr if r["kind"] == "sparse"]
    dn = [r for r in tr if r["kind"] == "dense"]
    assert len(sp) == 21
    bf16 = torch.bfloat16
    worst = 0.0

    # ---- sparse backbone: same execution order as SpMiddleResNetFHD.run_fused
    bb = net.backbone
    coors = _dev(np.pad(c, ((0, 0), (1, 0)))).int().contiguous()
    idx = bb.build_indexes(lambda i0: i0.mark(coors), 1, [int(g) for g in _grid_of(cfg.voxel_generator)], coors.device)
    perm = {}

    def to_hip(level, sparse_t, cpad):
        o = sparse_t.indices.numpy()
        if level not in perm:
            perm[level] = _rows_by_coord(idx[level].coords.cpu().numpy(), o)
        f = sparse_t.features.numpy()[perm[level]]
        out = np.zeros((f.shape[0], cpad), np.float32)
        out[:, : f.shape[1]] = f
        return _dev(out).to(bf16)

    layers = []
    for lvl, (conv, bn, blocks, is_subm_in) in enumerate(bb._stages()):
        src = lvl if is_subm_in else lvl - 1
        layers.append((conv, bn, src, lvl, False))
        for blk in blocks:
            layers.append((blk.conv1, blk.bn1, lvl, lvl, False))
            layers.append((blk.conv2, blk.bn2, lvl, lvl, True))
    assert len(layers) == 21
    for li, ((conv, bn, src, dst, has_res), rec) in enumerate(zip(layers, sp)):
        wpk, bias, cin_p, cout_p = conv.packed_weight(bf16, bn)
        ks, st, pd = conv.geometry()
        nbr = idx[src].rulebook(idx[dst], ks, st, pd)
        x_in = to_hip(src, rec["x"], cin_p)
        res = to_hip(dst, rec["residual"], cout_p) if has_res else None
        assert (rec["residual"] is not None) == has_res
        y = hip_ops.spconv_apply(x_in, wpk, bias, nbr, idx[dst].n, cout_p, residual=res, relu=True)
        want = to_hip(dst, rec["y"], cout_p).float().cpu().numpy()
        e = rel_err(y.float().cpu().numpy(), want)
        worst = max(worst, e)
        assert e <= BF16_ULP, "sparse layer %d (%d -> %d channels, level %d -> %d): %.3e" % (li, cin_p, cout_p, src, dst, e)
    report("%s bf16 teacher-forced: 21 sparse layers, worst element" % variant, worst, BF16_ULP)

    # ---- dense layers: the conv plan's own objects, NHWC bf16
    nhwc = lambda t: _dev(t.numpy()).to(bf16).permute(0, 2, 3, 1).contiguous()  # noqa: E731
    back = lambda t: t.float().permute(0, 3, 1, 2).cpu().numpy()  # noqa: E731
    rp, hp = RPNPlan(net.neck, bf16), HeadPlan(net.bbox_head, bf16)
    worst_d, k = 0.0, 0

    def check(name, got, ref):
        nonlocal worst_d
        e = rel_err(got, ref.numpy())
        worst_d = max(worst_d, e)
        assert e <= BF16_ULP, "%s: %.3e" % (name, e)

    for i, stack in enumerate(rp.blocks):
        for j, conv in enumerate(stack):
            check("rpn block %d conv %d" % (i, j), back(conv(nhwc(dn[k]["x"]))), dn[k]["y"])
            k += 1
        jd = i - rp.start
        if jd >= 0:
            kind, kk, op, cout = rp.deblocks[jd]
            x_in = nhwc(dn[k]["x"])
            B, H, W, _ = x_in.shape
            if kind == "conv":
                y = op(x_in)
            elif kind == "up":
                y = torch.empty((B, H * kk, W * kk, cout), dtype=bf16, device="cuda")
                for sub, dy, dx in op:
                    sub(x_in, out=y, co_off=0, osy=kk, osx=kk, ooy=dy, oox=dx)
            else:
                raise AssertionError("deblock kind %s is not part of the bf16 CenterPoint plan" % kind)
            check("rpn deblock %d" % jd, back(y), dn[k]["y"])
            k += 1
    for conv in hp.shared:
        check("head shared conv", back(conv(nhwc(dn[k]["x"]))), dn[k]["y"])
        k += 1
    for ti, (c1, c2, names, couts) in enumerate(hp.tasks):
        if hp.ff:
            p0, p1 = hp.pre[ti]
            x_in = nhwc(dn[k]["x"])
            if ti == 0:  # the plan's first forecast conv of task 0 reads [x | zeros] through zero weights
                x_in = torch.cat([x_in, torch.zeros_like(x_in)], dim=-1)
            check("head task %d forecast conv 0" % ti, back(p0(x_in)), dn[k]["y"])
            check("head task %d forecast conv 1" % ti, back(p1(nhwc(dn[k + 1]["x"]))), dn[k + 1]["y"])
            k += 2
        # oracle order per head name: first conv, final conv; the plan fuses all first convs (c1) and all final convs (c2)
        firsts, finals = dn[k: k + 2 * len(names): 2], dn[k + 1: k + 2 * len(names): 2]
        k += 2 * len(names)
        y1 = back(c1(nhwc(firsts[0]["x"])))
        hc = firsts[0]["y"].shape[1]
        for hi, name in enumerate(names):
            check("head task %d %s first conv" % (ti, name), y1[:, hi * hc:(hi + 1) * hc], firsts[hi]["y"])
        y2 = back(c2(nhwc(torch.cat([f["y"] for f in firsts], dim=1))))
        o = 0
        for hi, name in enumerate(names):
            check("head task %d %s final conv" % (ti, name), y2[:, o:o + couts[hi]], finals[hi]["y"])
            o += couts[hi]
    assert k == len(dn), (k, len(dn))
    report("%s bf16 teacher-forced: %d dense layers, worst element" % (variant, len(dn)), worst_d, BF16_ULP)


def test_bf16_decode_on_identical_maps(hip):
    """The decode + rotated NMS of the bf16 configuration on IDENTICAL head maps: the bf16 oracle's maps (300k-point cloud, n3) go
    through the device's predict and must give the oracle's detections at the fp32 criterion (rows within 1e-3, differences attributed)."""
    from futuredet_amd.synth import synthetic_cloud
    from oracle import bf16 as obf

    cfg, net, onet = _build_pair("forecast_n3")
    cloud = synthetic_cloud(seed=0, target_points=300000)
    _, _, _, ex = _bf16_example(cfg, cloud)
    _, _, opreds, odet = obf.run(onet, ex, cfg.test_cfg)
    preds = [{k: _dev(v.numpy()) for k, v in p.items()} for p in opreds]
    with torch.no_grad():
        got = net.bbox_head.predict({"metadata": [None]}, preds, cfg.test_cfg)[0]
    sc = torch.sigmoid(opreds[0]["hm"].float()).flatten()
    k = int(cfg.test_cfg["nms"]["nms_pre_max_size"])
    topk_cut = float(torch.topk(sc, k).values[-1]) if int((sc > cfg.test_cfg["score_threshold"]).sum()) > k else None
    assert len(odet[0]["scores"]) > 100
    _attribute("bf16 maps of the oracle through the device decode (config 3)", _rows(got), _rows(odet[0]), cfg.test_cfg, topk_cut=topk_cut)


def test_voxelnet_bf16_end_to_end_vs_fp32_oracle(hip):
    """BASELINE config 3 precision at the 30k-point size: bf16 conv features/weights with fp32 accumulation (sparse convs
    on the bf16 MFMA kernels, RPN + head on the hand-written NHWC conv plan, replayed as a hipGraph) against the fp32 CPU
    oracle."""
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, onet = _build_pair("forecast_n3")
    net.set_precision(torch.bfloat16)
    cloud = synthetic_cloud(seed=0, target_points=30000)
    with torch.no_grad():
        for _ in range(2):  # second call replays the captured graph
            got = net.forward_points([_dev(cloud)], cfg.voxel_generator, padded=False)[0]
    _check_bf16("30k n3", cfg, net, onet, cloud, got)


# ------------------------------------------------------------------------------------------------ full size (BASELINE configs 2-5)
def test_full_size_config2_fp32_vs_oracle(hip):
    """BASELINE configs[1] (the bench workload: forecast_n0, seed-0 300k-point cloud, fp32): backbone BEV map and neck
    output element-wise |d| <= 1e-3 * max(1, |ref|) vs the oracle, detections matched with attribution."""
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, onet = _build_pair("forecast_n0")
    cloud = synthetic_cloud(seed=0, target_points=300000)
    v, c, n, obb, obev, want = _oracle_run(cfg, onet, cloud)
    assert len(n) > 150000
    bb, x = _hip_maps(net, cfg, v, c, n)
    assert_close("full size config 2 (n0 fp32 300k) backbone BEV", bb.float().cpu().numpy(), obb.numpy(), 1e-3)
    assert_close("full size config 2 (n0 fp32 300k) neck output", x.float().cpu().numpy(), obev.numpy(), 1e-3)
    with torch.no_grad():
        for i in range(2):  # eager + graph replay
            got = net.forward_points([_dev(cloud)], cfg.voxel_generator, padded=False)[0]
            _attribute("full size config 2 forward_points run %d" % i, _rows(got), _rows(want), cfg.test_cfg, topk_cut=want["topk_cut"])


def test_full_size_config3_bf16_vs_oracle(hip):
    """BASELINE configs[2]: forecast_n3 (7-timestep head), 300k-point cloud, bf16, against the fp32 oracle."""
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, onet = _build_pair("forecast_n3")
    net.set_precision(torch.bfloat16)
    cloud = synthetic_cloud(seed=0, target_points=300000)
    with torch.no_grad():
        for _ in range(2):
            got = net.forward_points([_dev(cloud)], cfg.voxel_generator, padded=False)[0]
    _check_bf16("full size config 3 (n3 300k)", cfg, net, onet, cloud, got)


def test_config4_per_rank_batch8_bf16(hip):
    """BASELINE configs[3] per-rank workload: forecast_n3, bf16, 8 clouds of 300k points in one batch.  Samples are
    independent, so the batched result must equal the 8 single-sample runs (attributed matching; bit equality is
    recorded), and one sample is checked against the fp32 oracle like config 3."""
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, onet = _build_pair("forecast_n3")
    net.set_precision(torch.bfloat16)
    clouds = [synthetic_cloud(seed=s, target_points=300000) for s in range(8)]
    dev_clouds = [_dev(cl) for cl in clouds]
    with torch.no_grad():
        for _ in range(2):
            batch = net.forward_points(dev_clouds, cfg.voxel_generator, padded=False)
        bitwise = 0
        for b in range(8):
            solo = net.forward_points([dev_clouds[b]], cfg.voxel_generator, padded=False)[0]
            g, w = _rows(batch[b]), _rows(solo)
            bitwise += int(g.shape == w.shape and np.array_equal(g, w))
            _attribute("config 4 batch-of-8 sample %d vs its single-sample run" % b, g, w, cfg.test_cfg)
    report("config 4 samples bit-identical to single-sample runs", float(8 - bitwise), 8.0, "(count of samples that differ in any bit)")
    _check_bf16("config 4 sample 3 of the batch", cfg, net, onet, clouds[3], batch[3])


def test_full_size_config5_pedestrian_fine_grid_fp32_vs_oracle(hip):
    """BASELINE configs[4] single-GPU load: pedestrian forecast_n3, 500k-point cloud, 0.05 m grid (2160 x 2160 x 40,
    BEV 270 x 270), 400k-voxel cap.  Voxelizer and level-0 rulebook (pair count per tap) bit-exact vs the oracle, BEV
    maps element-wise 1e-3, detections attributed; the run is repeated to check determinism at this size."""
    from futuredet_amd.synth import synthetic_cloud
    from futuredet_amd.voxelize import points_to_voxel
    from oracle import ops as oops

    cfg, net, onet = _build_pair("forecast_n3", "pedestrian", voxel_size=(0.05, 0.05, 0.2), max_voxel_num=(300000, 400000))
    vg = cfg.voxel_generator
    cloud = synthetic_cloud(seed=0, target_points=500000)
    v, c, n, obb, obev, want = _oracle_run(cfg, onet, cloud)
    hv, hc, hn = points_to_voxel(cloud, vg["voxel_size"], vg["range"], vg["max_points_in_voxel"], True, vg["max_voxel_num"][1])
    assert np.array_equal(hc, c) and np.array_equal(hn, n) and np.array_equal(hv, v)
    report("config 5 voxelizer (500k pts, %d voxels) vs oracle" % len(n), 0.0, 0.0, "(bit-exact)")
    # level-0 SubM rulebook: same number of pairs per tap as the spconv-1.0 restatement
    D, H, W = 41, int(_grid_of(vg)[1]), int(_grid_of(vg)[0])
    idx4 = np.pad(c, ((0, 0), (1, 0))).astype(np.int32)
    _, _, pnum, _ = oops.rulebook(idx4, (D, H, W), (3, 3, 3), (1, 1, 1), (1, 1, 1), True)
    src = hip.SparseIndex(1, D, H, W, torch.device("cuda"))
    n_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    src.mark(_dev(idx4))
    src.scan(n_dev)
    src.finalize(int(n_dev.item()))
    nbr = src.rulebook(src, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    assert src.n == len(n)
    assert np.array_equal((nbr[:, :src.n] >= 0).sum(1).cpu().numpy(), np.asarray(pnum)), "pairs per tap differ from the oracle"
    bb, x = _hip_maps(net, cfg, v, c, n)
    assert_close("full size config 5 (ped n3 fp32 500k, 0.05 m) backbone BEV", bb.float().cpu().numpy(), obb.numpy(), 1e-3)
    assert_close("full size config 5 (ped n3 fp32 500k, 0.05 m) neck output", x.float().cpu().numpy(), obev.numpy(), 1e-3)
    with torch.no_grad():
        outs = [net.forward_points([_dev(cloud)], vg, padded=True) for _ in range(2)]
        got = net.forward_points([_dev(cloud)], vg, padded=False)[0]
    for a, b in zip(*outs):
        assert torch.equal(a, b), "forward_points must be deterministic"
    _attribute("full size config 5 forward_points", _rows(got), _rows(want), cfg.test_cfg, topk_cut=want["topk_cut"])
    # BASELINE.md runs this config in bf16 (conv features / weights, fp32 accumulate): same cloud against the fp32 oracle
    net.set_precision(torch.bfloat16)
    with torch.no_grad():
        for _ in range(2):
            got16 = net.forward_points([_dev(cloud)], vg, padded=False)[0]
    _check_bf16("full size config 5 (ped n3 500k, 0.05 m)", cfg, net, onet, cloud, got16)


def test_two_ranks_on_one_gpu_equal_single_process(hip, tmp_path):
    """The N > 1 path end to end on a 1-GPU box: bench.py under torch.distributed.run with 2 ranks that both use cuda:0
    (FD_BENCH_ONE_DEVICE=1, collectives over gloo): rank-strided shards of a global batch of 4 clouds, per-step fixed-shape
    all_gather, rank-0 JSON line.  The gathered, re-interleaved detections must equal a single-process run of the same 4
    samples, and the line must carry the multi-rank fields."""
    import json
    import subprocess
    import sys

    from futuredet_amd import dist_infer
    from futuredet_amd.synth import synthetic_cloud

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = os.path.join(str(tmp_path), "gathered.npz")
    env = dict(os.environ, FD_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--variant", "forecast_n3", "--points", "30000", "--batch", "2", "--global-batch", "4", "--no-cpu-baseline", "--no-host-leg",
           "--dump", dump]
    out = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2 and line["value"] > 0
    g = np.load(dump)
    got = dist_infer.unpack_results(torch.from_numpy(g["packed"]), torch.from_numpy(g["counts"]))
    assert len(got) == 4
    cfg, net, _ = _build_pair("forecast_n3")
    with torch.no_grad():
        for sid in range(4):  # global sample id = cloud seed (rank r computed samples r, r + 2)
            want = net.forward_points([_dev(synthetic_cloud(seed=sid, target_points=30000))], cfg.voxel_generator, padded=False)[0]
            _attribute("2 ranks on one GPU: gathered sample %d vs single process" % sid, _rows(got[sid]), _rows(want), cfg.test_cfg)


def test_two_ranks_weak_scaling_gathers_once_after_the_run(hip, tmp_path):
    """The default (weak-scaling) N > 1 mode on a 1-GPU box: each of 2 ranks runs its own stream of sweeps (pool seeds
    rank * pool + slot) as whole-sweep hipGraphs, results are exchanged by ONE all_gather after the last step (the reference's
    tools/dist_test.py:236-237); the dump holds the last step's detections of both ranks, rank-interleaved."""
    import json
    import subprocess
    import sys

    from futuredet_amd import dist_infer
    from futuredet_amd.synth import synthetic_cloud

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = os.path.join(str(tmp_path), "gathered_weak.npz")
    env = dict(os.environ, FD_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--points", "30000", "--pool", "2", "--no-cpu-baseline", "--no-host-leg", "--dump", dump]
    out = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 3
    assert "after the last step" in line["config"]["parallelism"] and "hipGraph" in line["config"]["workload"]
    g = np.load(dump)
    got = dist_infer.unpack_results(torch.from_numpy(g["packed"]), torch.from_numpy(g["counts"]))
    B = len(got) // 2  # clouds per rank and pass (the bench's default: two)
    assert B >= 1 and len(got) == 2 * B
    cfg, net, _ = _build_pair("forecast_n0")
    with torch.no_grad():
        for r in range(2):  # last step (index 2) uses pool slot 2 % 2 = 0 -> seeds (r * pool + 0) * B + b; gathered sample b * W + r
            for b in range(B):
                seed = (r * 2 + 0) * B + b
                want = net.forward_points([_dev(synthetic_cloud(seed=seed, target_points=30000))], cfg.voxel_generator, padded=False)[0]
                _attribute("2 ranks weak scaling: rank %d cloud %d of the last step vs single process" % (r, b), _rows(got[b * 2 + r]), _rows(want),
                           cfg.test_cfg)


def test_weights_reload_invalidates_captured_graph(hip):
    """The neck+head hipGraph and the folded / packed weight caches must not survive a weight change: run (graph
    captured), load other weights, run again -- the result must match the oracle with the NEW weights and differ from the
    old result; then the same after an in-place update without load_state_dict (caches are keyed on parameter versions)."""
    from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims

    cfg, net, onet = _build_pair("forecast_n0", seed=7)
    cloud = synthetic_cloud(seed=1, target_points=20000)
    with torch.no_grad():
        for _ in range(2):
            first = _rows(net.forward_points([_dev(cloud)], cfg.voxel_generator, padded=False)[0])
    # other weights in every stage (a different seed saturates every score to 1.0, which leaves the NMS order undefined)
    sd = tame_box_dims(seeded_state_dict(net, 7))
    for key, f in (("backbone.conv_input.0.weight", 0.8), ("neck.blocks.0.1.weight", 0.7), ("bbox_head.shared_conv.0.weight", 0.6),
                   ("bbox_head.tasks.0.hm.0.weight", 0.9)):
        sd[key] = sd[key] * f
    net.load_state_dict(sd, strict=False)
    onet.load_state_dict(sd, strict=False)
    want = _rows(_oracle_run(cfg, onet, cloud)[5])
    with torch.no_grad():
        got = _rows(net.forward_points([_dev(cloud)], cfg.voxel_generator, padded=False)[0])
    assert _match_detections(first, want) > 0.5 * len(want), "the weight change must change the detections for this test to mean anything"
    _attribute("after load_state_dict (fp32 graph)", got, want, cfg.test_cfg)
    # in-place parameter update (no load_state_dict)
    with torch.no_grad():
        net.bbox_head.shared_conv[0].weight.mul_(0.5)
        net.neck.blocks[0][4].weight.mul_(1.25)
        onet.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()}, strict=False)
        want2 = _rows(_oracle_run(cfg, onet, cloud)[5])
        got2 = _rows(net.forward_points([_dev(cloud)], cfg.voxel_generator, padded=False)[0])
    assert _match_detections(got, want2) > 0.5 * len(want2)
    _attribute("after an in-place weight update (fp32 graph)", got2, want2, cfg.test_cfg)


def test_full_size_backbone_is_deterministic_and_linear(hip):
    """Size-independent properties at the bench size (300k-point cloud, all 21 sparse convs): the HIP backbone has no
    atomics and a fixed summation order, so two runs agree bit for bit; and with BatchNorm folded and ReLU removed from
    a single SubM conv the op is linear: conv(a*x + y) == a*conv(x) + conv(y) within fp32 rounding."""
    from futuredet_amd import build_detector
    from futuredet_amd.configs import centerpoint_config
    from futuredet_amd.synth import seeded_state_dict, synthetic_cloud

    cfg = centerpoint_config("forecast_n0")
    net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    net.load_state_dict(seeded_state_dict(net, 7), strict=False)
    net = net.cuda().eval()
    cloud = [_dev(synthetic_cloud(seed=0, target_points=300000))]
    outs = []
    for _ in range(2):
        b, s, l, c = net.forward_points(cloud, cfg.voxel_generator, padded=True)
        outs.append((b.clone(), s.clone(), l.clone(), c.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    # linearity of one full-size 64->64 SubM convolution (level-2 geometry taken from the same cloud)
    bb = net.backbone
    vg = cfg.voxel_generator
    vox = hip.voxelize(cloud[0], vg["voxel_size"], vg["range"], 10, vg["max_voxel_num"][1], want_voxels=False, want_mean=True,
                       mean_stride=16, coor_cols=4)
    idx = bb.build_indexes(None, 1, [1440, 1440, 40], torch.device("cuda"), voxels=(vox["coors"], vox["num_voxels"], vg["max_voxel_num"][1]))
    i2 = idx[2]
    nbr = i2.rulebook(i2, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    g = torch.Generator("cuda").manual_seed(3)
    w = torch.randn((27, 64, 64), device="cuda", generator=g) * 0.05
    wpk = hip.pack_spconv_weight(w)
    x = torch.randn((i2.n, 64), device="cuda", generator=g)
    y = torch.randn((i2.n, 64), device="cuda", generator=g)
    f = lambda t: hip.spconv_apply(t, wpk, None, nbr, i2.n, 64)  # noqa: E731
    lhs, rhs = f(2.5 * x + y), 2.5 * f(x) + f(y)
    assert i2.n > 100000
    assert float((lhs - rhs).abs().max()) <= 1e-4 * float(rhs.abs().max())


# ------------------------------------------------------------------------------------------------ whole sweep as one hipGraph
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_static_step_graph_is_bit_identical_to_the_eager_sweep(hip, precision):
    """StaticStep (no host read-back between the voxelizer and the NMS output, the whole sweep replayed as ONE hipGraph with
    capacity-sized sparse levels and device-side counts) against forward_points on the same clouds: one captured graph, clouds
    of different sizes (one larger and two much smaller than the warm-up cloud, and an empty one) -- padded outputs and counts
    must be bit-identical, and the level counts the graph reports must equal the eager host counts."""
    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, _ = _build_pair("forecast_n3" if precision == "bf16" else "forecast_n0")
    if precision == "bf16":
        net.set_precision(torch.bfloat16)
    step = StaticStep(net, cfg.voxel_generator, capacity=90000)
    clouds = [_dev(synthetic_cloud(seed=s, target_points=n)) for s, n in ((3, 40000), (4, 80000), (5, 9000), (6, 300))]
    clouds.append(torch.zeros((0, 5), device="cuda"))
    with torch.no_grad():
        step.warm_up([clouds[0]])
        for i, c in enumerate(clouds):
            assert c.shape[0] <= 90000
            want = net.forward_points([c], cfg.voxel_generator)
            want_levels = list(net.last_level_counts)
            got = step([c])
            torch.cuda.synchronize()
            assert step.graph is not None
            cnt_w, cnt_g = want[3].cpu().numpy(), got[3].cpu().numpy()
            assert np.array_equal(cnt_w, cnt_g), (i, cnt_w, cnt_g)
            assert step.level_counts.cpu().tolist() == want_levels, (i, step.level_counts.cpu().tolist(), want_levels)
            for b in range(cnt_w.shape[0]):
                for s_ in range(cnt_w.shape[1]):
                    k = int(cnt_w[b, s_])
                    for name, w, g in (("boxes", want[0], got[0]), ("scores", want[1], got[1]), ("labels", want[2], got[2])):
                        assert torch.equal(w[b, s_, :k], g[b, s_, :k]), "cloud %d: %s of (sample %d, step %d) differ" % (i, name, b, s_)
            report("static step (%s) cloud %d (%d pts): detections bit-identical to eager" % (precision, i, c.shape[0]), 0.0, 0.0,
                   "(levels %s)" % want_levels)
    with pytest.raises(ValueError):
        step([torch.zeros((90001, 5), device="cuda")])


def test_static_step_follows_weight_updates(hip):
    """A captured whole-sweep graph must not outlive the weights it was captured with."""
    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, _ = _build_pair("forecast_n0", seed=7)
    cloud = _dev(synthetic_cloud(seed=1, target_points=20000))
    step = StaticStep(net, cfg.voxel_generator, capacity=32768)
    with torch.no_grad():
        first = [t.clone() for t in step([cloud])]
        g0 = step.graph
        net.bbox_head.shared_conv[0].weight.mul_(0.5)
        want = net.forward_points([cloud], cfg.voxel_generator)
        got = step([cloud])
        torch.cuda.synchronize()
    assert step.graph is not g0
    assert torch.equal(want[3], got[3]) and torch.equal(want[1], got[1]) and torch.equal(want[0], got[0])
    assert not torch.equal(first[1], got[1])


def test_static_step_full_size_matches_eager(hip):
    """The bench workload (forecast_n0, 300k-point cloud, fp32) through the whole-sweep graph with the real row capacities
    (level 1: 1.28 M rows, level 2: 1.43 M rows for <= 160k voxels) against the eager sweep, and a second, different cloud
    through the same graph."""
    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, _ = _build_pair("forecast_n0")
    clouds = [_dev(synthetic_cloud(seed=s, target_points=300000)) for s in (0, 1)]
    step = StaticStep(net, cfg.voxel_generator, capacity=330000)
    with torch.no_grad():
        step.warm_up([clouds[0]])
        for i, c in enumerate(clouds):
            want = net.forward_points([c], cfg.voxel_generator)
            levels = list(net.last_level_counts)
            got = step([c])
            torch.cuda.synchronize()
            assert step.level_counts.cpu().tolist() == levels
            assert levels[0] > 150000
            assert torch.equal(want[3], got[3])
            k = want[3].max().item()
            assert k > 0 and torch.equal(want[0][:, :, :k], got[0][:, :, :k]) and torch.equal(want[1][:, :, :k], got[1][:, :, :k])
            report("static step full size cloud %d: bit-identical to eager" % i, 0.0, 0.0, "(levels %s)" % levels)


def test_static_step_at_the_voxel_cap(hip):
    """Level 0 exactly full: with max_voxel_num = 3000 a 40k-point cloud hits the voxelizer's cap, so the device count of
    level 0 equals its row capacity (the boundary of every min(capacity, count) in the capacity launches)."""
    import copy

    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, _ = _build_pair("forecast_n0")
    vg = copy.deepcopy(dict(cfg.voxel_generator))
    vg["max_voxel_num"] = [3000, 3000]
    cloud = _dev(synthetic_cloud(seed=3, target_points=40000))
    step = StaticStep(net, vg, capacity=45056)
    with torch.no_grad():
        want = net.forward_points([cloud], vg)
        levels = list(net.last_level_counts)
        assert levels[0] == 3000
        for _ in range(2):
            got = step([cloud])
            torch.cuda.synchronize()
            assert step.level_counts.cpu().tolist() == levels
            assert torch.equal(want[3], got[3]) and torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])


def test_static_step_pointpillars(hip):
    """The PointPillars configs through StaticStep (voxelizer with point slots -> pillar reader -> scatter -> RPN -> head ->
    decode as one hipGraph on a fixed-capacity cloud buffer): bit-identical to forward_points for clouds of several sizes."""
    from futuredet_amd import build_detector
    from futuredet_amd.configs import pointpillars_config
    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims

    cfg = pointpillars_config("car")
    net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    net.load_state_dict(tame_box_dims(seeded_state_dict(net, 7)), strict=False)
    net = net.cuda().eval()
    step = StaticStep(net, cfg.voxel_generator, capacity=65536)
    with torch.no_grad():
        for i, (seed, n) in enumerate(((3, 30000), (4, 60000), (5, 2000))):
            c = _dev(synthetic_cloud(seed=seed, target_points=n))
            want = net.forward_points([c], cfg.voxel_generator)
            got = step([c])
            torch.cuda.synchronize()
            assert torch.equal(want[3], got[3]) and torch.equal(want[0], got[0]) and torch.equal(want[1], got[1]) and torch.equal(want[2], got[2])
            assert int(want[3].sum()) > 0
            report("static step PointPillars cloud %d: bit-identical to eager" % i, 0.0, 0.0, "(%d pillars)" % int(step.level_counts.cpu()[0]))


def test_static_step_batch_of_two_random_sizes(hip):
    """StaticStep with two clouds per sweep (batch column, per-sample voxel counts on the device) over a sequence of random
    cloud sizes, including an empty sample next to a full one: every replay equals the eager sweep bit for bit."""
    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, _ = _build_pair("forecast_n3")
    rng = np.random.default_rng(5)
    step = StaticStep(net, cfg.voxel_generator, capacity=49152, batch_size=2)
    sizes = [(20000, 30000), (45000, 0), (300, 41000), (0, 0), (33000, 33000)] + [tuple(int(v) for v in rng.integers(0, 44000, 2)) for _ in range(5)]
    with torch.no_grad():
        for i, (na, nb) in enumerate(sizes):
            clouds = []
            for j, n in enumerate((na, nb)):
                c = _dev(synthetic_cloud(seed=100 + 2 * i + j, target_points=max(n, 300)))
                clouds.append(c[:n] if n < 300 else c[:min(c.shape[0], 49152)])
            want = net.forward_points(clouds, cfg.voxel_generator)
            levels = list(net.last_level_counts)
            got = step(clouds)
            torch.cuda.synchronize()
            assert step.level_counts.cpu().tolist() == levels, (i, step.level_counts.cpu().tolist(), levels)
            assert torch.equal(want[3], got[3]), (i, want[3], got[3])
            cw = want[3].cpu().numpy()
            for b in range(cw.shape[0]):
                for s_ in range(cw.shape[1]):
                    k = int(cw[b, s_])
                    assert torch.equal(want[0][b, s_, :k], got[0][b, s_, :k]) and torch.equal(want[1][b, s_, :k], got[1][b, s_, :k])
    report("static step B=2, %d random size pairs: bit-identical to eager" % len(sizes), 0.0, 0.0)


def test_static_steps_in_flight_on_two_streams_do_not_share_head_buffers(hip):
    """the forecast_feature head (n3dtf) lays its concat out in two ping-pong buffers.  They must belong to a
    call / a captured graph, not to the plan: two StaticSteps on different streams, fed DIFFERENT clouds and replayed
    concurrently many times, must each reproduce their serial result bit for bit."""
    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, _ = _build_pair("forecast_n3dtf")
    clouds = [_dev(synthetic_cloud(seed=s, target_points=n)) for s, n in ((11, 60000), (12, 25000))]
    streams = [torch.cuda.Stream() for _ in clouds]
    steps = [StaticStep(net, cfg.voxel_generator, capacity=65536) for _ in clouds]
    with torch.no_grad():
        want = [[t.clone() for t in net.forward_points([c], cfg.voxel_generator)] for c in clouds]
        for st, s, c in zip(steps, streams, clouds):  # warm-up + capture, serially
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                st.warm_up([c])
                st([c])
            s.synchronize()
        for rep in range(12):
            outs = []
            for st, s, c in zip(steps, streams, clouds):
                with torch.cuda.stream(s):
                    outs.append([t.clone() for t in st([c])])
            torch.cuda.synchronize()
            for i, (w, g) in enumerate(zip(want, outs)):
                assert torch.equal(w[3], g[3]), (rep, i)
                k = int(w[3].max())
                assert k > 0 and torch.equal(w[0][:, :, :k], g[0][:, :, :k]) and torch.equal(w[1][:, :, :k], g[1][:, :, :k]), (rep, i)
    report("n3dtf: two whole-sweep graphs in flight on two streams, 12 rounds: each bit-identical to its serial result", 0.0, 0.0)


def test_dense_conv_tiles_bit_identical(hip):
    """The fp32 conv plan tunes only the workgroup tile of a layer's fixed formulation: every tile shape of
    the direct kernel, and every tile shape of the Winograd kernel, must give the same bits (the per-element summation order
    does not depend on the tile), so a timing-dependent tile choice cannot change results between runs or ranks."""
    rng = np.random.default_rng(3)
    for (ks, stride, cin, cout, H, W) in ((3, 1, 64, 128, 37, 77), (3, 2, 32, 128, 40, 33), (1, 1, 128, 256, 23, 18), (3, 1, 128, 70, 30, 26), (3, 1, 32, 200, 9, 63)):
        x = torch.from_numpy(rng.standard_normal((2, H, W, cin)).astype(np.float32)).cuda()
        w = torch.from_numpy((rng.standard_normal((cout, cin, ks, ks)) * (2.0 / (cin * ks * ks)) ** 0.5).astype(np.float32))
        b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
        wpk = hip.pack_conv2d_weight_f32(w).cuda()
        ref, n_ok = None, 0
        for tile in range(0, hip.conv2d_f32_num_tiles() + 1):
            try:
                y = hip.conv2d_nhwc_f32(x, wpk, b, cout, ks, stride, True, tile=tile)
            except hip.FutureDetHipError:
                continue  # (pointwise variants refuse a 3x3 layer and vice versa)
            ref = y if ref is None else ref
            assert torch.equal(ref, y), "direct k%d s%d %d->%d: tile %d changes the result" % (ks, stride, cin, cout, tile)
            n_ok += 1
        assert n_ok >= 4
        if ks == 3 and stride == 1:
            wpw = hip.pack_conv2d_weight_wino(w).cuda()
            ref = None
            for tile in range(0, hip.conv2d_wino_f32_num_tiles() + 1):
                try:
                    y = hip.conv2d_wino_nhwc_f32(x, wpw, b, cout, True, tile=tile)
                except hip.FutureDetHipError:
                    assert tile == 7 and W < 63, (tile, W)  # the strip variant refuses images narrower than one strip of 32 tiles
                    continue
                ref = y if ref is None else ref
                assert torch.equal(ref, y), "winograd %d->%d: tile %d changes the result" % (cin, cout, tile)
    report("dense fp32 convs: all tile shapes of one formulation bit-identical", 0.0, 0.0)


def test_forecast_groups_beyond_one_box_per_thread(hip):
    """fd_forecast_groups with more boxes than the workgroup has threads (the reference's multi_future has no 1024-box
    limit; here 8192): connected components of the 'closer than match_thresh' graph, numbered by their smallest member,
    against scipy's connected_components on the float64 distance matrix.  Chains of up to ~60 boxes exercise the label sweeps."""
    from scipy.sparse.csgraph import connected_components

    rng = np.random.default_rng(9)
    for n in (1500, 3000):
        # clusters of random walks (long chains) plus isolated boxes
        pts = []
        while len(pts) < n:
            L = int(rng.integers(1, 60))
            p = rng.uniform(-50, 50, 3)
            for _ in range(L):
                pts.append(p.copy())
                p = p + rng.uniform(-0.12, 0.12, 3)
        c = np.asarray(pts[:n], np.float64)
        c = c[rng.permutation(n)]
        d = np.sqrt(np.maximum(((c[:, None] - c[None]) ** 2).sum(-1), 0.0))
        adj = d < 0.25
        _, lab = connected_components(adj, directed=False)
        first = {}
        for i, l in enumerate(lab):
            first.setdefault(l, i)
        order = {l: r for r, l in enumerate(sorted(first, key=first.get))}
        want = np.array([order[l] for l in lab], np.int32)
        # pairs within 1e-9 of the threshold could legitimately differ (distance formula): none in this draw
        assert not (np.abs(d - 0.25) < 1e-9).any()
        got = hip.forecast_groups(torch.from_numpy(c).cuda(), 0.25).cpu().numpy()
        assert np.array_equal(got, want), (n, int((got != want).sum()))
    report("forecast groups, 1500 / 3000 boxes: component ids equal scipy's", 0.0, 0.0)


def _run_bench_ranks(tmp_path, n, extra, name, timeout=1500):
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = os.path.join(str(tmp_path), name + ".npz")
    env = dict(os.environ, FD_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 30100 + (os.getpid() % 700)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", str(n), "--no-cpu-baseline", "--no-host-leg", "--dump", dump] + extra
    out = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    return line, np.load(dump)


def test_eight_ranks_on_one_gpu_weak_scaling(hip, tmp_path):
    """VERDICT r2 #7b: the driver's `--gpus 8` command shape with 8 ranks sharing cuda:0 (collectives over gloo): 8 processes, rank-local
    build check, rank-0 weight broadcast + checksum, per-rank CPU slices, whole-sweep graphs per rank, ONE gather after the last
    step.  The last step's detections of all 8 ranks equal single-process runs of the same seeds."""
    from futuredet_amd import dist_infer
    from futuredet_amd.synth import synthetic_cloud

    line, g = _run_bench_ranks(tmp_path, 8, ["--steps", "3", "--warmup", "1", "--points", "20000", "--pool", "2", "--inflight", "2"], "weak8")
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["steps"] == 3 and line["value"] > 0
    assert "checksum" in line["config"]["replicas"] and "all 8 rank" in line["config"]["replicas"]
    got = dist_infer.unpack_results(torch.from_numpy(g["packed"]), torch.from_numpy(g["counts"]))
    B = len(got) // 8  # clouds per rank and pass (the bench's default: two)
    assert B >= 1 and len(got) == 8 * B
    cfg, net, _ = _build_pair("forecast_n0")
    with torch.no_grad():
        for r in range(8):  # last step (index 2) uses pool slot 0 -> seeds (r * pool + 0) * B + b; gathered sample b * W + r
            for b in range(B):
                want = net.forward_points([_dev(synthetic_cloud(seed=r * 2 * B + b, target_points=20000))], cfg.voxel_generator, padded=False)[0]
                _attribute("8 ranks weak scaling: rank %d cloud %d of the last step vs single process" % (r, b), _rows(got[b * 8 + r]), _rows(want),
                           cfg.test_cfg)


def test_eight_ranks_on_one_gpu_config4_strong_scaling(hip, tmp_path):
    """BASELINE configs[3] in the shape the driver would launch it (`--config 4 --gpus 8`: forecast_n3 bf16, global batch 64 = seeds
    0..63 split rank-strided, micro-batches of 4 per rank, per-step gather), with smaller clouds so that eight processes share one
    GPU: the 64 gathered samples come back in global order and equal a single-process bf16 run of the same seeds."""
    from futuredet_amd import dist_infer
    from futuredet_amd.synth import synthetic_cloud

    line, g = _run_bench_ranks(tmp_path, 8, ["--config", "4", "--points", "15000", "--steps", "1", "--warmup", "1", "--inflight", "1"], "strong8")
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["dtype"] == "bf16" and "global batch 64 over 8" in line["config"]["workload"]
    got = dist_infer.unpack_results(torch.from_numpy(g["packed"]), torch.from_numpy(g["counts"]))
    assert len(got) == 64
    cfg, net, _ = _build_pair("forecast_n3")
    net.set_precision(torch.bfloat16)
    with torch.no_grad():
        for sid in (0, 1, 7, 8, 9, 31, 63):  # rank sid % 8, local sample sid // 8
            want = net.forward_points([_dev(synthetic_cloud(seed=sid, target_points=15000))], cfg.voxel_generator, padded=False)[0]
            a, b = _rows(got[sid]), _rows(want)
            assert a.shape == b.shape and np.array_equal(a, b), "sample %d of the 8-rank run differs from the single-process bf16 run" % sid


def test_plain_bench_command_starts_its_own_ranks(hip):
    """`python bench.py --gpus 2` WITHOUT a launcher (no WORLD_SIZE in the environment) must run two ranks -- it
    re-executes itself under torch.distributed.run on a free port (tools/dist_test.py:125-135 is started once per GPU by the
    launcher) -- and report n_gpus = 2; with fewer devices than ranks (and no one-device hook) it must refuse, not print n_gpus 1."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FD_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--points", "20000", "--no-cpu-baseline",
           "--no-host-leg", "--inflight", "2", "--pool", "2"]
    out = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0 and "all 2 rank" in line["config"]["replicas"]
    if torch.cuda.device_count() < 2:
        env.pop("FD_BENCH_ONE_DEVICE")
        out = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "device(s) are visible" in out.stderr and "n_gpus" not in out.stdout


@pytest.mark.parametrize("variant,precision", [("forecast_n0", "fp32"), ("forecast_n3", "bf16"), ("forecast_n3dtf", "fp32"), ("forecast_n3dtf", "bf16")])
def test_decode_reads_the_head_output_in_place(hip, variant, precision):
    """predict_packed (the decode kernels read the conv plan's NHWC output through fd_map_view, velocities / labels / counts are
    assembled by fd_assemble_detections: seven launches, no torch kernel) against the generic path on the same maps (NCHW float
    copies per map + torch index_select / gather / cat): identical detections, bit for bit, for the standard head (one task,
    velocity channels per step) and the dense forecast_feature head (one task per step), fp32 and bf16."""
    from futuredet_amd.dist_infer import pack_results
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, _ = _build_pair(variant)
    if precision == "bf16":
        net.set_precision(torch.bfloat16)
    clouds = [synthetic_cloud(seed=s, target_points=25000) for s in (21, 22)]
    with torch.no_grad():
        for B in (1, 2):
            exs = [_bf16_example(cfg, cl) for cl in clouds[:B]]
            maps = []
            for v, c, n, _ in exs:
                maps.append(_hip_maps(net, cfg, v, c, n)[1])
            x = torch.cat(maps, 0)
            preds = net.bbox_head(x)
            assert getattr(preds[0], "raw", None) is not None, "the plan's head output must carry its buffer"
            packed, counts = net.bbox_head.predict_packed(preds, cfg.test_cfg)
            plain = [dict(pd) for pd in preds]  # plain dicts: no buffer attached -> the generic path
            assert net.bbox_head.predict_packed(plain, cfg.test_cfg) is None
            want_p, want_c = pack_results(*net.bbox_head.predict_padded(plain, cfg.test_cfg))
            assert torch.equal(counts, want_c) and int(counts.sum()) > 0
            assert torch.equal(packed, want_p), "%s %s B=%d: fused decode output differs from the generic path" % (variant, precision, B)
    report("%s %s: decode on the in-place head output == generic path (B = 1, 2)" % (variant, precision), 0.0, 0.0)


def test_static_step_high_water_mark_capacities_and_overflow(hip):
    """sparse levels of a captured sweep sized from a high-water mark (1.5 x the warm-up cloud's row counts) instead
    of the data-free bounds.  A sweep within the capacities is bit-identical to the eager path; a cloud that needs MORE rows is
    detected from the level counts (overflowed()), run_checked() returns the eager result for it, nothing is written outside the
    tables (the next small sweep through the same graph is still exact); and an fp32 batch of 8 -- whose data-free capacities
    exceed the fp32 kernel's 2^23-row packing limit, so round 2 refused to capture it -- is captured and exact."""
    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, _ = _build_pair("forecast_n0")
    vg = cfg.voxel_generator
    small, big = _dev(synthetic_cloud(seed=1, target_points=15000)), _dev(synthetic_cloud(seed=2, target_points=85000))

    def same(a, b):
        k = int(a[3].max())
        return torch.equal(a[3], b[3]) and torch.equal(a[0][:, :, :k], b[0][:, :, :k]) and torch.equal(a[1][:, :, :k], b[1][:, :, :k])

    with torch.no_grad():
        step = StaticStep(net, vg, capacity=90112, row_caps="auto", headroom=1.5)
        step.warm_up([small])
        got = [t.clone() for t in step([small])]
        torch.cuda.synchronize()
        free = StaticStep(net, vg, capacity=90112, row_caps="datafree")  # data-free capacities, for the comparison of sizes
        free.expected = step.expected
        free.capture()
        assert free.caps is None and step.caps is not None and step.caps[1] < 0.2 * 8 * 160000, step.caps
        assert not step.overflowed()
        want_small = net.forward_points([small], vg)
        assert same(want_small, got)
        got_big = [t.clone() for t in step([big])]  # (default: checked -- detected and re-run on the eager path)
        torch.cuda.synchronize()
        assert step.overflowed(), (step.level_counts.cpu().tolist(), step.caps)
        want_big = net.forward_points([big], vg)
        assert same(want_big, got_big), "step(clouds) must return the eager result when the replay overflowed"
        assert same(want_big, step.run_checked([big]))
        raw = step([big], check=False)  # the unchecked replay: wrong rows, which is why the caller has to look at overflowed()
        torch.cuda.synchronize()
        assert step.overflowed() and raw is step.outputs
        again = step([small])
        torch.cuda.synchronize()
        assert not step.overflowed() and same(want_small, again), "an overflowing sweep must not damage the captured step"
        # fp32, 8 clouds per sweep
        clouds = [_dev(synthetic_cloud(seed=30 + i, target_points=12000 + 1500 * i)) for i in range(8)]
        with pytest.raises(NotImplementedError):
            s8 = StaticStep(net, vg, capacity=24576, batch_size=8, row_caps="datafree")
            s8.warm_up(clouds)
            s8.capture()
        s8 = StaticStep(net, vg, capacity=24576, batch_size=8, row_caps="auto")
        s8.warm_up(clouds)
        got8 = s8(clouds)
        torch.cuda.synchronize()
        assert not s8.overflowed() and same(net.forward_points(clouds, vg), got8)
    report("static step with high-water-mark capacities: exact within, detected + eager beyond, fp32 batch of 8 captured", 0.0, 0.0,
           "(caps %s)" % step.caps)


def test_static_step_with_bev_map_head(hip):
    """VERDICT r2 missing #4: the forecast_n3dtfm head (bev_map branch, center_head.py:336-341,380-381) through the whole-sweep
    graph: the rasterised map is a static input of the captured step; replays with different clouds AND different maps equal the
    eager sweep bit for bit, and the map matters (another map changes the detections)."""
    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import synthetic_cloud

    cfg, net, _ = _build_pair("forecast_n3dtfm")
    rng = np.random.default_rng(4)
    maps = [_dev(rng.uniform(0, 1, (1, 6, 180, 180)).astype(np.float32)) for _ in range(3)]
    clouds = [_dev(synthetic_cloud(seed=s, target_points=n)) for s, n in ((40, 30000), (41, 18000), (42, 42000))]
    step = StaticStep(net, cfg.voxel_generator, capacity=49152, row_caps="auto", headroom=3.0)
    outs = []
    with torch.no_grad():
        step.warm_up([clouds[0]], bev_map=maps[0])
        for c, m in zip(clouds, maps):
            want = net.forward_points([c], cfg.voxel_generator, bev_map=m)
            got = step([c], bev_map=m)
            torch.cuda.synchronize()
            assert step.graph is not None and not step.overflowed()
            k = int(want[3].max())
            assert k > 0 and torch.equal(want[3], got[3]) and torch.equal(want[0][:, :, :k], got[0][:, :, :k]) and torch.equal(want[1][:, :, :k], got[1][:, :, :k])
            outs.append(got[1].clone())
        other = step([clouds[2]], bev_map=maps[0])
        torch.cuda.synchronize()
        assert not torch.equal(other[1], outs[2]), "the bev_map input must reach the head"
    report("static step with a bev_map head (n3dtfm): 3 clouds x 3 maps bit-identical to eager", 0.0, 0.0)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_street_scene_with_few_detections(hip, precision):
    """a scene of typical nuScenes sparsity next to the saturated one -- synthetic_cloud(profile="street"): 300k points
    in ~65k voxels (motion-compensated static scene), heat-map head tamed (synth.tame_scores) so that a few hundred cells pass the
    score threshold and a few dozen boxes survive: neither the 160k-voxel cap, nor the top-1000 cut, nor the 83-per-step cut is
    active.  fp32: maps 1e-3 and detections attributed against the oracle; bf16: against the bf16 oracle (distribution +
    attribution).  The cloud also goes through a whole-sweep graph that was warmed up on a DENSE cloud (launch grids sized by
    typical counts of another density) and must equal the eager sweep bit for bit."""
    from futuredet_amd.detectors import StaticStep
    from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims, tame_scores

    cfg, net, onet = _build_pair("forecast_n3")
    sd = tame_scores(tame_box_dims(seeded_state_dict(net, 7)))
    net.load_state_dict(sd, strict=False)
    onet.load_state_dict(sd, strict=False)
    cloud = synthetic_cloud(seed=0, target_points=300000, profile="street")
    if precision == "bf16":
        net.set_precision(torch.bfloat16)
    with torch.no_grad():
        got = net.forward_points([_dev(cloud)], cfg.voxel_generator, padded=False)[0]
    n_det = len(got["scores"])
    assert 8 <= n_det < 300, n_det
    if precision == "fp32":
        v, c, n, obb, obev, want = _oracle_run(cfg, onet, cloud)
        assert 40000 < len(n) < 90000 and want["topk_cut"] is None
        bb, x = _hip_maps(net, cfg, v, c, n)
        assert_close("street scene (n3 fp32, %d voxels) backbone BEV" % len(n), bb.float().cpu().numpy(), obb.numpy(), 1e-3)
        assert_close("street scene (n3 fp32) neck output", x.float().cpu().numpy(), obev.numpy(), 1e-3)
        _attribute("street scene (n3 fp32, %d detections)" % n_det, _rows(got), _rows(want), cfg.test_cfg)
    else:
        _check_bf16("street scene (n3 bf16, %d detections)" % n_det, cfg, net, onet, cloud, got, min_rows=8)
    dense = _dev(synthetic_cloud(seed=1, target_points=300000))
    step = StaticStep(net, cfg.voxel_generator, capacity=330000, row_caps="auto")
    with torch.no_grad():
        step.warm_up([dense])
        want_p = net.forward_points([_dev(cloud)], cfg.voxel_generator)
        got_p = step([_dev(cloud)])
        torch.cuda.synchronize()
    assert not step.overflowed()
    k = int(want_p[3].max())
    assert torch.equal(want_p[3], got_p[3]) and torch.equal(want_p[0][:, :, :k], got_p[0][:, :, :k]) and torch.equal(want_p[1][:, :, :k], got_p[1][:, :, :k])


# ------------------------------------------------------------------------------------------------ FutureDet end to end (8f-2 + A15 + 8f-3)
def _random_packed(rng, B, T, post, counts):
    packed = np.zeros((B, T, post, 11), np.float32)
    for b in range(B):
        base = rng.uniform(-40, 40, (post, 2))
        for t in range(T):
            n = int(counts[b, t])
            xy = base[rng.permutation(post)[:n]] + rng.normal(0, 0.4, (n, 2)) + 0.3 * t
            packed[b, t, :n, :2] = xy
            packed[b, t, :n, 2] = rng.normal(-1, 0.3, n)
            packed[b, t, :n, 3:6] = rng.uniform(0.5, 5, (n, 3))
            packed[b, t, :n, 6:8] = rng.normal(0, 3, (n, 2))
            packed[b, t, :n, 8] = rng.uniform(-4, 4, n)
            packed[b, t, :n, 9] = rng.uniform(0.1, 1, n)
            packed[b, t, :n, 10] = t
    return packed


@pytest.mark.gpu
def test_forecast_from_detections_equals_the_single_sweep_calls_and_the_oracle(hip):
    """fd_forecast_from_detections (one call per batch, everything of a sample in device memory) against (a) the three single-sweep entry
    points it batches -- fd_det_to_global_boxes, fd_forecast_chains, fd_forecast_groups, each pinned by the reference's goldens above --
    bit for bit, and (b) the CPU restatement (oracle/forecast.py: tracker's trajectory list and order, multi_future's ids).  Cases: full
    steps, ragged steps, a sample with an empty step (tracker returns no trajectory), duplicated first boxes (one forecast group)."""
    from futuredet_amd import forecast, hip_ops
    from oracle import forecast as oforecast

    rng = np.random.default_rng(5)
    B, T, post = 4, 7, 83
    counts = np.array([[83] * T, list(rng.integers(20, 83, T)), [30, 31, 0, 29, 30, 31, 32], [40] * T], np.int32)
    packed = _random_packed(rng, B, T, post, counts)
    packed[3, 0, 1, :3] = packed[3, 0, 0, :3] + np.float32(0.05)   # two first boxes closer than 0.25 m: one forecast id
    time = rng.uniform(0.4, 0.6, (B, T - 1))
    rec = np.zeros((B, 14))
    for b in range(B):
        q1, q2 = rng.normal(0, 1, 4), rng.normal(0, 1, 4)
        rec[b] = np.concatenate([q1 / np.linalg.norm(q1) * (1.0 if b else 1.0 + 1e-9), rng.normal(0, 2, 3), q2 / np.linalg.norm(q2), rng.normal(0, 300, 3)])
    out = forecast.sweep_forecast(_dev(packed), _dev(counts), _dev(time), _dev(rec), classname="car")
    out = forecast.sweep_forecast(_dev(packed), _dev(counts), _dev(time), _dev(rec), classname="car", out=out)  # (buffers are reused)
    h = out.host()
    n_traj_total = 0
    for b in range(B):
        cs, pose = (rec[b, :4], rec[b, 4:7]), (rec[b, 7:11], rec[b, 11:14])
        c, q, v, s = hip_ops.det_to_global_boxes(_dev(packed[b].reshape(-1, 11)[:, :9].copy()), cs, pose)
        for name, t_ in (("center", c), ("quat", q), ("velocity", v), ("size", s)):
            assert np.array_equal(h[name][b].reshape(t_.shape), t_.cpu().numpy()), (b, name)
        single = hip_ops.forecast_chains(c.reshape(T, post, 3), v.reshape(T, post, 3), _dev(counts[b]), _dev(time[b]), 2.0)
        for name in ("fwd_idx", "fwd_ok", "bwd_idx", "bwd_ok", "match_idx", "cv_centers"):
            assert np.array_equal(h[name][b], single[name].cpu().numpy()), (b, name)
        assert int(h["status"][b]) == int(single["status"].cpu()[0]) == int((counts[b] == 0).any())
        # the oracle's tracker on the same global boxes: trajectory list, order, centres
        cen = [h["center"][b, t, :counts[b, t]] for t in range(T)]
        vel = [h["velocity"][b, t, :counts[b, t]] for t in range(T)]
        want = oforecast.tracker("car", list(time[b]), cen, vel)
        got = forecast.trajectories_from_arrays(h, b)
        if want is None:
            assert got == [] and int(h["n_traj"][b]) == 0 and (h["traj_kind"][b] == -1).all()
            continue
        fwd, cv, bwd = want
        assert [k for k, _, _, _ in got] == [0] * len(fwd) + [1] * len(cv) + [2] * len(bwd), b
        for (kind, gid, centres, idx), ref in zip(got, list(fwd) + [None] * len(cv) + list(bwd)):
            if ref is not None:
                assert list(idx) == list(ref)
        for j, i in enumerate(range(len(fwd), len(fwd) + len(cv))):
            assert np.array_equal(got[i][2], cv[j]), (b, j)
        firsts = np.stack([tr[2][0] for tr in got])
        ids = oforecast.forecast_ids(firsts)
        assert np.array_equal(np.array([tr[1] for tr in got]), ids), b
        assert np.array_equal(hip_ops.forecast_groups(_dev(firsts), 0.25).cpu().numpy(), ids)
        if b == 3:
            assert got[len(fwd)][1] == got[len(fwd) + 1][1], "the two near-identical first boxes share a forecast id"
        n_traj_total += len(got)
    report("forecast from detections: %d sweeps, %d trajectories, boxes / chains / ids identical to the single-sweep calls and the oracle" % (B, n_traj_total), 0.0, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("variant,precision", [("forecast_n0", "fp32"), ("forecast_n3", "bf16")])
def test_full_sweep_step_equals_assembly_plus_sweep_plus_forecast(hip, variant, precision):
    """FullSweepStep -- raw sweeps + transforms -> fd_sweep_assemble -> whole sweep -> fd_forecast_from_detections as ONE captured
    graph -- against the same three stages run one by one (eager assembly of the same raw rows, forward_points on the assembled cloud,
    the forecast call on its detections): detections and every forecast array bit-identical, for two clouds per pass, replayed on clouds
    other than the one it was captured with (different raw row counts, different descriptors)."""
    from futuredet_amd import forecast, hip_ops
    from futuredet_amd.detectors import FullSweepStep
    from futuredet_amd.synth import synthetic_sweeps

    cfg, net, _ = _build_pair(variant)
    if precision == "bf16":
        net.set_precision(torch.bfloat16)

    def sample(seed, n):
        raw, rows, mats, lags, close = synthetic_sweeps(seed=seed, target_points=n)
        desc = hip_ops.sweep_descriptors(rows, mats, lags, close)
        rng = np.random.default_rng(seed)
        q1, q2 = rng.normal(0, 1, 4), rng.normal(0, 1, 4)
        rec = np.concatenate([q1 / np.linalg.norm(q1), rng.normal(0, 2, 3), q2 / np.linalg.norm(q2), rng.normal(0, 300, 3)])
        return dict(raw=_dev(raw), desc=torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).cuda(), time=_dev(rng.uniform(0.4, 0.6, 6)),
                    records=_dev(rec), host_desc=desc)

    samples = [sample(s, n) for s, n in ((11, 40000), (12, 60000), (13, 25000), (14, 52000))]
    step = FullSweepStep(net, cfg.voxel_generator, capacity=70000, n_sweeps=10, batch_size=2)
    with torch.no_grad():
        step.warm_up(samples[:2])
        for pair in (samples[:2], samples[2:], [samples[3], samples[0]]):
            packed, counts = [t.clone() for t in step(pair)]
            got = step.forecast.host()
            clouds = []
            for smp in pair:
                pts, cnt = hip_ops.assemble_sweeps(smp["raw"], smp["host_desc"])
                clouds.append(pts[: int(cnt.cpu()[0])].contiguous())
            want_p, want_c = net.forward_points(clouds, cfg.voxel_generator, padded="packed")
            assert torch.equal(want_c, counts)
            for b in range(2):
                for t in range(counts.shape[1]):
                    k = int(counts[b, t])
                    assert torch.equal(want_p[b, t, :k], packed[b, t, :k]), (b, t)
            want = forecast.sweep_forecast(packed, counts, torch.stack([s["time"] for s in pair]), torch.stack([s["records"] for s in pair]), "car").host()
            for name in want:
                assert np.array_equal(want[name], got[name], equal_nan=True), name
            assert int(counts.sum()) > 0 and int(got["n_traj"].sum()) > 0
    report("full sweep step (%s %s): detections and forecast arrays bit-identical to assembly + sweep + forecast run one by one" % (variant, precision), 0.0, 0.0,
           "(%d detections, %d trajectories in the last pass)" % (int(counts.sum()), int(got["n_traj"].sum())))
