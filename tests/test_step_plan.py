"""Host side of the step engine without a GPU: the ctypes mirror of ``gpsa_step_desc`` against the C struct
(``gpsa_step_describe`` builds the same plan ``gpsa_step_create`` does, on the host only, and reports what it
derived), argument validation, and the arena sizes' behaviour."""
import ctypes as C

import pytest

from spatial_alignment_amd import _lib


def describe(V=2, D=2, S=5, mx=200, mg=200, mods=((50, 50, 0, (10000, 10000)),), fixed=None, s_test=0, n_test=None,
             want_kl=1, kw=0, kd=0, kl_own=None):
    lib = _lib.load()
    d = _lib.StepDesc()
    if kl_own is not None:
        d.kl_own_lo, d.kl_own_hi = kl_own
    d.n_views, d.n_dims, d.n_mods, d.n_samples = V, D, len(mods), S
    d.m_x, d.m_g, d.kind_warp, d.kind_data = mx, mg, kw, kd
    rows = []
    for i, (L, P, lmc, r) in enumerate(mods):
        d.n_latent[i], d.n_out[i], d.has_lmc[i], d.n_rows[i] = L, P, lmc, sum(r)
        d.n_test[i] = 0 if n_test is None else n_test[i]
        rows += list(r)
    d.s_test, d.want_kl = s_test, want_kl
    fx = (C.c_int * V)(*(fixed or [0] * V))
    rw = (C.c_longlong * len(rows))(*rows)
    d.view_fixed, d.view_rows = fx, rw
    out = (C.c_longlong * 7)()
    rc = lib.gpsa_step_describe(C.byref(d), out)
    return rc, list(out)[:6], int(out[6])


def test_headline_plan():
    rc, (saved, scratch, n_kl, eps, runs, Cs), nokeep = describe()
    assert rc == 0
    assert n_kl == 2 * 2 + 50                      # Omega_G rows + the data GP's outputs
    assert eps == 2 * 5 * 10000 * 2                # S draws [n_v, D] per free view
    assert runs == 1 and Cs == 10048               # both views in one batch; 64-column granularity
    # saved: alpha fp32 [M, S N] + Sigma [L, S N] + the warp GPs' fp64 alpha and D kept products + the batch
    C_ = 5 * 20000
    floor = 200 * C_ * 4 + 50 * C_ * 4 + 2 * 200 * Cs * 8 * (1 + 2) + 2 * 57 * 200 * 200 * 8
    # + what a fused-ELBO forward leaves for its backward instead of kept products: g [L+1, S N], dmean [L, S N], abar [M, S N]
    floor += (2 * 50 + 1 + 200) * C_ * 4
    assert floor <= nokeep <= 1.15 * floor
    # a training forward also keeps the data GP's products Omega_l alpha behind everything else: [L, M, S N] fp32,
    # padded to the kernel's tiles (208 rows, 192-column tiles)
    assert 0 <= saved - nokeep - 50 * 208 * 100032 * 4 <= 4096
    assert scratch >= 200 * C_ * 8                  # the fp64 covariance panel of the data GP is in there


def test_fixed_views_split_runs_and_kl_terms():
    rc, (_, _, n_kl, eps, runs, Cs), _nk = describe(V=5, S=2, mx=30, mg=20, mods=((7, 7, 0, (10, 20, 30, 40, 50)),),
                                              fixed=[0, 0, 1, 0, 0])
    assert rc == 0
    assert n_kl == 5 * 2 + 7                       # fixed views keep their (absent) terms' slots
    assert eps == 2 * 2 * (10 + 20 + 40 + 50)
    assert runs == 2 and Cs == 64                  # views {0,1} and {3,4}: a fixed view breaks the uniform strides


def test_many_views_are_batched_by_sixteen():
    rc, out, _nk = describe(V=40, S=1, mx=16, mg=16, mods=((3, 3, 0, tuple([8] * 40)),))
    assert rc == 0 and out[4] == 3 and out[2] == 40 * 2 + 3


def test_two_modalities_lmc_and_test_pass():
    rc, (saved, scratch, n_kl, eps, runs, Cs), _nk = describe(
        V=2, D=3, S=3, mx=16, mg=18, mods=((7, 7, 0, (64, 64)), (2, 4, 1, (36, 36))), s_test=1, n_test=(17, 5))
    assert rc == 0 and n_kl == 2 * 3 + 7 + 2 and eps == 3 * 3 * (100 + 100) and Cs == 128
    rc2, (saved2, scratch2, *_r), _nk2 = describe(
        V=2, D=3, S=3, mx=16, mg=18, mods=((7, 7, 0, (64, 64)), (2, 4, 1, (36, 36))))
    assert saved > saved2                          # the test passes keep their own alpha / Sigma


def test_sizes_grow_with_the_sample_count():
    a = describe(S=1)[1]
    b = describe(S=5)[1]
    assert b[0] > a[0] and b[1] > a[1] and b[3] == 5 * a[3]


@pytest.mark.parametrize("kw", [dict(D=5), dict(D=0), dict(mx=0), dict(mods=((50, 40, 0, (10, 10)),)),
                                dict(mods=((50, 50, 0, (10, -1)),)), dict(V=0, mods=((5, 5, 0, ()),))])
def test_invalid_descriptions_are_refused(kw):
    assert describe(**kw)[0] == _lib.GPSA_EINVAL


def test_row_total_must_match():
    lib = _lib.load()
    d = _lib.StepDesc()
    d.n_views, d.n_dims, d.n_mods, d.n_samples, d.m_x, d.m_g = 2, 2, 1, 1, 8, 8
    d.n_latent[0], d.n_out[0], d.n_rows[0], d.want_kl = 3, 3, 99, 1
    fx, rw = (C.c_int * 2)(0, 0), (C.c_longlong * 2)(10, 20)
    d.view_fixed, d.view_rows = fx, rw
    out = (C.c_longlong * 7)()
    assert lib.gpsa_step_describe(C.byref(d), out) == _lib.GPSA_EINVAL
    assert lib.gpsa_step_describe(C.byref(d), None) == _lib.GPSA_EINVAL


def test_view_rows_memo_sees_in_place_mutation():
    """the memo of 'views are consecutive row blocks' must not outlive the index objects' CONTENT (the engine
    path ignores view_idx, so a stale answer would assign rows to the wrong views)."""
    import types

    import numpy as np
    import torch

    from spatial_alignment_amd import step_engine as SE

    model = types.SimpleNamespace(n_views=2, modality_names=["m"])
    Ns = {"m": 10}
    # ndarrays (what create_view_idx_dict returns)
    vi = {"m": [np.arange(0, 4), np.arange(4, 10)]}
    assert SE.view_rows(model, vi, Ns) == (4, 6)
    assert SE.view_rows(model, vi, Ns) == (4, 6)            # the memo's hit
    vi["m"][0][1], vi["m"][0][2] = 2, 1                     # same object, same length, another order
    assert SE.view_rows(model, vi, Ns) is None
    vi["m"][0][1], vi["m"][0][2] = 1, 2
    assert SE.view_rows(model, vi, Ns) == (4, 6)
    # tensors: the version counter is part of the key
    vt = {"m": [torch.arange(0, 4), torch.arange(4, 10)]}
    assert SE.view_rows(model, vt, Ns) == (4, 6)
    assert SE.view_rows(model, vt, Ns) == (4, 6)
    vt["m"][1][[0, 1]] = vt["m"][1][[1, 0]]
    assert SE.view_rows(model, vt, Ns) is None
    # a write through another view of the same storage is seen too
    base = torch.arange(0, 10)
    vb = {"m": [base[:4], base[4:]]}
    assert SE.view_rows(model, vb, Ns) == (4, 6)
    base[5] = 7
    assert SE.view_rows(model, vb, Ns) is None
    # lists are never memoised
    vl = {"m": [list(range(0, 4)), list(range(4, 10))]}
    assert SE.view_rows(model, vl, Ns) == (4, 6)
    vl["m"][0][0] = 3
    assert SE.view_rows(model, vl, Ns) is None


def test_owner_computes_range_leaves_the_arena_layout_alone():
    """gpsa_step_desc.kl_own_lo / _hi (a data-parallel rank evaluates its own range of the KL terms): the batch keeps
    every matrix's slot - the data GP's pass reads all the Omega_l -, only the factorisation and the KL kernels skip"""
    base = describe()
    for own in ((0, 7), (7, 14), (48, 54), (54, 54)):
        got = describe(kl_own=own)
        assert got[0] == 0 and got[1] == base[1] and got[2] == base[2], own
