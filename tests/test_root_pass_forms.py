"""The light tree's root pass in two algebraic forms (csrc/device/dev_light.h tree_prepass): the reference's reservoir update per child and lane
(cuda/ris.cuh:138-148: accept = r < p; r = accept ? r / p : (r - p) / (1 - p)) and the fast flavour's threshold form (LUM_ROOT_THRESHOLD: accept = t W_c > 1 with
t = (1 - r0) / W0 constant between two accepted children). CPU, numpy: in binary64 the two forms pick the same child every time; in binary32 each of them
departs from real arithmetic about equally often - the scan is an expanding map (an accepted child multiplies the rounding error by 1 / p), which is also why a
build that rounds a reciprocal differently already picks differently - so neither is 'the' binary32 answer outside the exact flavour's fixed operation order."""
import numpy as np

HI32 = np.frombuffer(np.uint32(0x3F7FFFFF).tobytes(), np.float32)[0]


def reference_form(w, u, dt, recip_ulps=0):
    r, total, pick = dt(u), dt(0), -1
    for c, wc in enumerate(w.astype(dt)):
        total = dt(total + wc)
        if not wc > 0:
            continue
        p = dt(wc / total)
        ia, ir = dt(dt(1) / p), (dt(dt(1) / dt(dt(1) - p)) if p < 1 else dt(np.inf))
        if recip_ulps:  # a hardware reciprocal: the last bit may differ
            ia = np.nextafter(ia, dt(np.inf) if recip_ulps > 0 else dt(0), dtype=dt)
            ir = np.nextafter(ir, dt(0) if recip_ulps > 0 else dt(np.inf), dtype=dt)
        if r < p:
            pick, r = c, dt(r * ia)
        else:
            r = dt(dt(r - p) * ir)
        assert r >= 0, "the clamp's lower bound never acts (clamp_random_top)"
        r = min(r, dt(HI32) if dt == np.float32 else dt(1 - 2.0 ** -53))
    return pick


def threshold_form(w, u, dt):
    total, pick, t = dt(0), -1, None
    for c, wc in enumerate(w.astype(dt)):
        if not wc > 0:
            continue
        before, total = total, dt(total + wc)
        if before == 0:
            t, pick = dt(dt(dt(1) - dt(u)) / wc), c
            continue
        step = dt(before / dt(wc * total))
        excess = dt(np.float64(t) * np.float64(total) - 1.0) if dt == np.float32 else t * total - 1  # one rounding: the fma
        if excess > 0:
            t, pick = dt(excess * step), c
    return pick


def test_the_two_forms_are_the_same_function_in_real_arithmetic():
    rng = np.random.RandomState(1)
    for _ in range(2500):
        n = rng.randint(1, 65)
        w = rng.gamma(0.5, 1.0, n) * (rng.rand(n) > 0.1)  # some children without importance, the first ones included
        u = rng.rand()
        assert reference_form(w, u, np.float64) == threshold_form(w, u, np.float64)


def test_in_binary32_both_forms_stray_from_real_arithmetic_equally_often():
    rng = np.random.RandomState(2)
    n_cases, ref_off, thr_off, rcp_off = 4000, 0, 0, 0
    for _ in range(n_cases):
        w = rng.gamma(0.5, 1.0, 64).astype(np.float32)
        u = np.float32(rng.rand())
        truth = reference_form(w.astype(np.float64), np.float64(u), np.float64)
        ref_off += reference_form(w, u, np.float32) != truth
        thr_off += threshold_form(w, u, np.float32) != truth
        rcp_off += reference_form(w, u, np.float32, recip_ulps=1) != truth
    ref_off, thr_off, rcp_off = ref_off / n_cases, thr_off / n_cases, rcp_off / n_cases
    assert 0.01 < ref_off < 0.12, ref_off   # the reference's own binary32 scan of 64 children ends elsewhere than real arithmetic's in a few per cent of all scans
    assert thr_off < 1.5 * ref_off + 0.01, (thr_off, ref_off)
    assert rcp_off > 0.5 * ref_off, (rcp_off, ref_off)  # ... and so does the same form with reciprocals one ulp off (the fast flavour's v_rcp_f32)
