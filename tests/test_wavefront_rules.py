"""pathIterateKernel's own path rules (Integrators/PseudoRealtime/WavefrontPathTracer.cu:51-164; tracer parameter PathSemantics = Wavefront, oracle/ocore.h
pathTraceWavefront) against PathTrace<DIRECT> (Integrators/PathTracer.cu:10-113; the default) — the difference between the two estimators, on the CPU:

  at equal MaxPathLength the wavefront rules take no next-event estimation at the last vertex and cast fewer rays; their expectation is PathTrace's minus
  exactly that term — i.e. it equals PathTrace with the last vertex' next-event estimation left out (the oracle's what-if `omit_last_nee`)."""
import numpy as np

from cudatracerlib_amd import scenes


def test_wavefront_rules_are_path_trace_minus_the_last_vertex_next_event_estimation(orc):
    sc = scenes.cornell_box(64, 64); d = sc.desc
    n = 96
    tables = orc.sequence_tables(n)
    def per_pass(**kw):
        out = []; rays = 0
        for k in range(n):
            img, r = orc.render(d, 64, 64, n_passes=1, tables=tables[k:k + 1], max_path_length=3, rr_start=50, threads=8, **kw)
            out.append(img[..., :3].sum() / img[..., 6].sum()); rays += r
        out = np.array(out)
        return out.mean(), out.std(ddof=1) / np.sqrt(n), rays
    pt, pt_se, pt_rays = per_pass()
    wf, wf_se, wf_rays = per_pass(wavefront_rules=True)
    cut, cut_se, cut_rays = per_pass(omit_last_nee=True)
    # the wavefront rules are darker than PathTrace by far more than the noise, and cast fewer rays ...
    assert pt - wf > 5 * np.hypot(pt_se, wf_se), (pt, wf, pt_se, wf_se)
    assert wf_rays < pt_rays
    # ... and agree with PathTrace-without-the-last-NEE (two different estimators of the same integral: different sample use, same expectation)
    assert abs(wf - cut) < 4 * np.hypot(wf_se, cut_se), (wf, cut, wf_se, cut_se)
    assert abs(wf - cut) < 0.02 * cut


def test_wavefront_rules_with_one_bounce_see_only_emitters(orc):
    """MaxPathLength = 1: pathDepth + 1 == maxPathDepth at the first vertex — emission is added, nothing is sampled (WavefrontPathTracer.cu:79, :111), one ray per path;
    PathTrace adds the first vertex' next-event estimation"""
    sc = scenes.cornell_box(32, 32); d = sc.desc
    t = orc.sequence_tables(1)
    wf, wf_rays = orc.render(d, 32, 32, n_passes=1, tables=t, max_path_length=1, wavefront_rules=True)
    pt, pt_rays = orc.render(d, 32, 32, n_passes=1, tables=t, max_path_length=1)
    assert wf_rays == 32 * 32 and pt_rays > wf_rays
    lit = wf[..., :3].sum(2) > 0
    assert 0 < lit.mean() < 0.2                     # only the pixels that look at the light
    assert (pt[..., :3].sum(2) > 0).mean() > 0.5       # (one light sample per pixel: about 40 % of them are shadowed or face away)


def test_sixteen_bit_barycentrics_move_the_image_slightly(orc):
    sc = scenes.cornell_box(32, 32, glass_sphere=True); d = sc.desc
    t = orc.sequence_tables(2)
    a, _ = orc.render(d, 32, 32, n_passes=2, tables=t, max_path_length=5, wavefront_rules=True)
    b, _ = orc.render(d, 32, 32, n_passes=2, tables=t, max_path_length=5, wavefront_rules=True, u16_barycentrics=True)
    assert not np.array_equal(a, b)
    assert abs(a[..., :3].mean() - b[..., :3].mean()) < 0.02 * a[..., :3].mean()
