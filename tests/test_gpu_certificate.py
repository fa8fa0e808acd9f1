"""The CERTIFICATE of the coarse passes, tested directly: for every (row, query) of a small shard the coarse
score the scan kernels really compute (`lynse_hip_flat_coarse_scores`: one emit-all stage of the real pipeline) must lie within the
per-query bound E of the reference-order f32 score (oracle) — the whole pipeline keeps candidates by `coarse >= tau - 2E`, so one
pair outside the bound could lose a true neighbour without any end-to-end test noticing.

Constructed worst cases (the bound's terms):  v_d = min_d + (code_d + eps_d) / scale_d,  w_d = q_d / scale_d = s_q (u_d +
eta_d);  q.v - coarse = s_q sum eta_d c'_d + sum w_d eps_d  with |eps|, |eta| <= 1/2 and c' = code - 128.  The target row has every
code fractional part at +0.4999 and the signed code at -128 (the row that realises A1 = max row L1 norm), the query every rounding
residual at -0.4999: all D error terms have the same sign and the achieved |error| / E is ~0.95 (reported, asserted <= 1 and > 0.9) —
for IP and for squared L2 on the plain codes, on f32 and on F16 shards.  The augmented-L2 and unit-row cosine forms and the f16
shadow are held against the bound on hostile random data plus the analogous aligned-rounding row (ratios reported).
"""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
f32 = np.float32
FORM_I8, FORM_AUG, FORM_L2N, FORM_COSQ = 1, 2, 4, 8


@pytest.fixture(scope="module")
def L():
    import lynsedb_amd as L_

    assert L_._lib.device_count() >= 1
    return L_


def aligned_worst_case(dim, n, rng, frac=0.4999, f16_rows=False):
    """Rows whose SQ8 fit is min 0, scale 64 in every dimension (anchor rows 0 and 1), a target row (2) with all fractional parts
    at +frac and all signed codes at -128, random rows with random fractional parts elsewhere; queries whose rounding residuals are
    all -frac (q 0), all +frac (q 1: aligned with the all-maximum anchor row's +127 codes) and random (the rest)."""
    delta = 1.0 / 64.0
    rows = np.empty((n, dim), np.float64)
    rows[0] = 0.0
    rows[1] = 255.0 * delta
    rows[2] = frac * delta
    codes = rng.integers(0, 256, (n - 3, dim)).astype(np.float64)
    rows[3:] = np.clip(codes + rng.uniform(-0.49, 0.49, (n - 3, dim)), 0.0, 255.0) * delta
    rows[3] = (255.0 - frac) * delta              # codes +127 with fractional parts at -frac (eps = -frac)
    data = rows.astype(f32)
    if f16_rows:
        data = data.astype(np.float16).astype(f32)
    s_q = 2.0 ** -10
    nq = 8
    q = np.empty((nq, dim), np.float64)
    m = rng.integers(60, 127, (nq, dim)).astype(np.float64)
    resid = rng.uniform(-0.49, 0.49, (nq, dim))
    resid[0] = -frac
    resid[1] = +frac
    q[:] = (m + resid) * s_q / delta
    q[:, 0] = 127.0 * s_q / delta                 # fixes s_q = max |w| / 127 exactly
    return data, q.astype(f32)


def exact_scores(orc, queries, data, metric, f16_rows):
    out = np.empty((queries.shape[0], data.shape[0]), np.float64)
    for qi in range(queries.shape[0]):
        if f16_rows:
            out[qi] = [orc.distance_f16(queries[qi], data[r], metric) for r in range(data.shape[0])] if data.shape[0] <= 64 else \
                np.asarray(orc.all_distances(queries[qi], data, metric), np.float64)
        else:
            out[qi] = np.asarray(orc.all_distances(queries[qi], data, metric), np.float64)
    return out


def check_bound(L, orc, data, queries, metric_name, metric, coarse, f16_rows=False, want_form=None):
    dim = data.shape[1]
    idx = L.FlatIndex(None, dim, dtype="f16") if f16_rows else L.FlatIndex(None, dim)
    if f16_rows:
        idx.write_f16_bits(data.astype(np.float16).view(np.uint16))
    else:
        idx.write(data)
    idx.finalize()
    scores, bound, form = idx.coarse_scores(queries, metric_name, coarse)
    if want_form is not None:
        assert form == want_form, (form, want_form)
    exact = exact_scores(orc, queries, data, metric, False)   # (f32 kernels; the f16 kernels' sequential sums differ by less than the bound's rounding term)
    err = np.abs(scores.astype(np.float64) - exact)
    ratio = err / bound.astype(np.float64)[:, None]
    worst = np.unravel_index(np.argmax(ratio), ratio.shape)
    assert np.all(np.isfinite(scores)) and np.all(bound > 0)
    assert ratio.max() <= 1.0, (metric_name, coarse, "f16 rows" if f16_rows else "f32 rows", float(ratio.max()), worst, float(err[worst]), float(bound[worst[0]]))
    return float(ratio.max()), worst


@pytest.mark.parametrize("f16_rows", [False, True])
@pytest.mark.parametrize("metric_name,metric,dim,want_form", [("ip", O.IP, 256, FORM_I8), ("l2", O.L2, 256, FORM_I8 | FORM_L2N)])
def test_int8_bound_is_tight_and_holds_on_the_aligned_worst_case(L, oracle, metric_name, metric, dim, want_form, f16_rows):
    rng = np.random.default_rng(11)
    data, queries = aligned_worst_case(dim, 4096, rng, frac=0.499 if f16_rows else 0.4999, f16_rows=f16_rows)
    ratio, worst = check_bound(L, oracle, data, queries, metric_name, metric, "i8", f16_rows, want_form)
    print(f"certificate int8 {metric_name} dim {dim} {'F16' if f16_rows else 'f32'} rows: max |coarse - exact| / E = {ratio:.4f} at (query, row) = {worst}")
    # the construction puts every term of the bound on one side: the bound is met to within its 2 % safety factor (and not by luck:
    # the worst pair is the constructed one)
    assert ratio > 0.9 and worst[1] in (1, 2, 3) and worst[0] in (0, 1), (ratio, worst)


@pytest.mark.parametrize("metric_name,metric,dim,want_form", [("l2", O.L2, 200, FORM_I8 | FORM_AUG), ("cosine", O.COS, 256, FORM_I8 | FORM_COSQ),
                                                             ("ip", O.IP, 100, FORM_I8), ("cosine", O.COS, 100, FORM_I8 | FORM_COSQ)])
def test_int8_bound_holds_for_the_augmented_and_unit_row_forms(L, oracle, metric_name, metric, dim, want_form):
    rng = np.random.default_rng(12)
    data, queries = aligned_worst_case(dim, 4096, rng)
    ratios = [check_bound(L, oracle, data, queries, metric_name, metric, "i8", False, want_form)[0]]
    # hostile random data: a constant dimension, a 1e-6-range dimension, lognormal row scales, mixed signs, an offset dimension
    data = rng.standard_normal((6000, dim)).astype(f32) * np.exp(rng.normal(0.0, 1.0, (6000, 1))).astype(f32)
    data[:, 1] = 3.0
    data[:, 2] = 1.0 + 1e-6 * rng.random(6000).astype(f32)
    data[:, 3] += 100.0
    queries = rng.standard_normal((16, dim)).astype(f32)
    queries[3] = data[17] * 1.0001
    ratios.append(check_bound(L, oracle, data, queries, metric_name, metric, "i8", False, want_form)[0])
    print(f"certificate int8 {metric_name} dim {dim}: max |coarse - exact| / E = {ratios[0]:.4f} (aligned rows), {ratios[1]:.4f} (hostile random)")


@pytest.mark.parametrize("f16_rows", [False, True])
@pytest.mark.parametrize("metric_name,metric", [("ip", O.IP), ("l2", O.L2), ("cosine", O.COS)])
def test_f16_shadow_bound_holds_with_aligned_roundings(L, oracle, metric_name, metric, f16_rows):
    """The f16 coarse pass rounds both operands to 11 significant bits: elements just BELOW a rounding midpoint on both sides
    (1 + 0.49 ulp) put 2 x 0.49 ulp of relative error with one sign on every product; the target row is parallel to the query and
    has the largest norm, so (2u + u^2) |q| max |v| is what it realises."""
    rng = np.random.default_rng(13)
    dim, n = 192, 4096
    ulp = 2.0 ** -10
    base = (1.0 + 0.49 * ulp)
    data = (rng.uniform(0.2, 0.9, (n, dim)) * rng.choice([-1.0, 1.0], (n, dim))).astype(f32)
    data[0] = base
    data[1] = -base
    queries = rng.standard_normal((8, dim)).astype(f32)
    queries[0] = base
    queries[1] = (1.0 - 0.49 * ulp / 2)           # just above a midpoint: the other sign
    if f16_rows:
        data = data.astype(np.float16).astype(f32)   # (an F16 shard's rows ARE f16 values: only the query side rounds)
    ratio, worst = check_bound(L, oracle, data, queries, metric_name, metric, "f16", f16_rows, 0)
    print(f"certificate f16 {metric_name} {'F16' if f16_rows else 'f32'} rows: max |coarse - exact| / E = {ratio:.4f} at {worst}")
    if metric == O.IP and not f16_rows:
        assert ratio > 0.5, ratio     # (the bound also carries the subnormal floor and the accumulation terms: not met as tightly as the int8 one)


# ---- the Cauchy-Schwarz forms of the two quantisation terms as the BINDING ones (VERDICT r5 "weak" 2) ---------------------------------
# E's terms are min(Hoelder, Cauchy-Schwarz) each (k_i8c_prep_queries): |sum w eps| <= ||w|| max_rows ||eps|| and
# |s_q sum eta c'| <= s_q ||eta|| max_rows ||c'||.  The aligned worst case above makes both forms coincide; here the residual vector is
# PARALLEL to its partner with varying magnitudes, so Cauchy-Schwarz holds with equality while Hoelder is ~1.5x slack — a wrong
# max ||eps|| / max ||c'|| statistic (not refreshed by an append, or without the f32 evaluation error of the residual) would put the
# constructed pair outside the bound.
def _cs_case(kind, dim, n, rng):
    """Rows with SQ8 fit min 0 / scale 64 in every dimension from SPREAD anchors (16 rows; each holds 0 and 255/64 in dim / 16 of the
    dimensions and the exact mid code elsewhere: small norms), filler rows near the mid code, and ONE target row.
      kind "eps_w": the target row's quantisation residuals are eps_d = 0.4999 u_d / 127 for query 0's integer image u (query 0 is exactly
                    representable: eta = 0); every other row keeps |eps| <= 0.2.
      kind "eta_c": every row sits on code points (eps = 0); the target row has the largest ||c'|| (codes spread over the whole range) and
                    query 0's rounding residuals are eta_d = 0.4999 c'_d / 127."""
    delta, s_q, R = 1.0 / 64.0, 2.0 ** -10, 16
    anchors = np.full((R, dim), 128.0)
    for d in range(dim):
        anchors[d % R, d] = 0.0
        anchors[(d + R // 2) % R, d] = 255.0
    filler_codes = rng.integers(100, 157, (n - R - 1, dim)).astype(np.float64)
    u = rng.integers(20, 127, dim).astype(np.float64) * rng.choice([-1.0, 1.0], dim)      # (|u_d| <= 126 beside u_0: no w_d + eta_d above 127)
    u[0] = 127.0                                                    # fixes s_q = max |w| / 127 = 2^-10 exactly
    if kind == "eps_w":
        filler = filler_codes + rng.uniform(-0.2, 0.2, filler_codes.shape)
        target = rng.integers(108, 149, dim).astype(np.float64) + 0.4999 * u / 127.0
        w0 = u                                                       # in units of s_q
    else:
        filler = filler_codes
        cprime = rng.integers(30, 128, dim).astype(np.float64) * rng.choice([-1.0, 1.0], dim)
        cprime[1] = -127.0
        target = cprime + 128.0
        w0 = u + 0.4999 * cprime / 127.0
    base = (np.vstack([anchors, filler]) * delta).astype(f32)
    nq = 6
    q = (rng.integers(40, 127, (nq, dim)) + rng.uniform(-0.49, 0.49, (nq, dim))) * rng.choice([-1.0, 1.0], (nq, dim))
    q[0] = w0
    q[:, 0] = 127.0
    queries = (q * s_q / delta).astype(f32)
    return base, (target * delta).astype(f32)[None, :], queries, u


@pytest.mark.parametrize("kind", ["eps_w", "eta_c"])
@pytest.mark.parametrize("appended", [False, True])
def test_cauchy_schwarz_term_is_binding_and_met_on_a_fresh_shard_and_after_an_append(L, oracle, kind, appended):
    rng = np.random.default_rng(21)
    dim, n = 128, 4096
    base, target, queries, u = _cs_case(kind, dim, n, rng)
    idx = L.FlatIndex(None, dim)
    if appended:      # the shard answers (and builds its codes + statistics) WITHOUT the target row first; the append must raise max ||eps|| / max ||c'||
        idx.write(base)
        idx.finalize()
        s0, b0, form0 = idx.coarse_scores(queries, "ip", "i8")
        assert form0 == FORM_I8
        ex0 = exact_scores(oracle, queries, base, O.IP, False)
        assert (np.abs(s0.astype(np.float64) - ex0) / b0.astype(np.float64)[:, None]).max() <= 1.0
        idx.write(target)
        idx.finalize()
    else:
        idx.write(np.vstack([base, target]))
        idx.finalize()
    data = np.vstack([base, target])
    scores, bound, form = idx.coarse_scores(queries, "ip", "i8")
    assert form == FORM_I8 and scores.shape == (queries.shape[0], n)
    exact = exact_scores(oracle, queries, data, O.IP, False)
    ratio = np.abs(scores.astype(np.float64) - exact) / bound.astype(np.float64)[:, None]
    worst = np.unravel_index(np.argmax(ratio), ratio.shape)
    # the two forms of the binding term for query 0, in real arithmetic (what k_i8c_prep_queries evaluates in f64)
    s_q = 2.0 ** -10
    if kind == "eps_w":
        cs = s_q * np.linalg.norm(u) * 0.4999 * np.linalg.norm(u) / 127.0          # ||w|| ||eps||: eps = 0.4999 u / 127
        hoelder = 0.5 * s_q * np.abs(u).sum()
    else:
        cp = (data[-1].astype(np.float64) * 64.0) - 128.0
        cs = s_q * (0.4999 * np.linalg.norm(cp) / 127.0) * np.linalg.norm(cp)      # s_q ||eta|| ||c'||: eta = 0.4999 c' / 127
        hoelder = 0.5 * s_q * np.abs(cp).sum()
    print(f"certificate C-S {kind} {'after an append' if appended else 'fresh shard'}: max |coarse - exact| / E = {ratio.max():.4f} at {worst}; "
          f"Cauchy-Schwarz term {cs:.4f} vs Hoelder {hoelder:.4f}; E(q0) = {float(bound[0]):.4f}")
    assert cs < 0.8 * hoelder                                   # Cauchy-Schwarz is the strictly smaller form here ...
    assert float(bound[0]) < 1.02 * hoelder                     # ... and the one E is made of (Hoelder alone would already exceed E)
    assert ratio.max() <= 1.0, (float(ratio.max()), worst)
    assert worst == (0, n - 1) and ratio.max() > 0.9, (float(ratio.max()), worst)      # met by the constructed pair, to within the 2 % safety factor + the f32 terms
    if appended:
        assert float(bound[0]) > 1.5 * float(b0[0]), (float(bound[0]), float(b0[0]))   # the append raised the statistic the bound is made of
