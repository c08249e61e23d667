"""Visibility rays visit the farthest child first (dev_trace.h ShadowState, LUM_SHADOW_ORDER): any occluder ends such a ray, and for a ray that
leaves a surface the nearest boxes - the surface it has just left and its neighbours - are the worst place to look for one. Results cannot depend
on the order (every parity test runs on it); what the order buys is node visits, and that is what this test holds on to: in a closed scene a
visibility ray must not cost more node visits than 0.85 of a closest-hit ray (nearest first it costs about as many: 15.2 against 16.4 on the
1.43 M-triangle hall, farthest first 11.4)."""
import pytest

from luminary_amd import scenes
from luminary_amd.core import Core

pytestmark = pytest.mark.gpu


def test_visibility_rays_need_fewer_node_visits_than_closest_hit_rays():
    host = scenes.hall_scene(320, 180, 6, target_triangles=200_000)
    core = Core(0)
    try:
        core.upload(host.device_scene())
        core.set_ambient_reuse(0)  # every visibility ray traced: the visiting order of ALL of them is what is measured
        core.set_pixels(None)
        core.reset_counters()
        core.render(0, 4, samples_per_pass=4)
        cnt = core.counters()
        closest, shadow = cnt[4] / max(cnt[0], 1), cnt[6] / max(cnt[1], 1)
        assert cnt[0] > 100_000 and cnt[1] > 100_000
        assert shadow < 0.85 * closest, "node visits per visibility ray %.2f, per closest-hit ray %.2f" % (shadow, closest)
    finally:
        core.close()
