// The visibility-ray kernel (optix/optix_kernel_shadow.cu:15-100, cuda/optix_anyhit.cuh:49-139) and its query type, in a header of their own because the
// fast flavour compiles the kernel in ITS OWN translation unit (csrc/device/wavefront_fast_shadow.hip) with another instruction scheduler:
// `-mllvm -amdgpu-sched-strategy=max-ilp` takes 3-4 % off k_shadow_rays (hall 180.2 -> 172.4 ms per 3 steps, scan 40.9 -> 39.8, Example-class 26.65 -> 25.8)
// and ADDS 2.4 % to k_shade and 0.8 % to k_trace when applied to the whole flavour (profiles/r05_ab_experiments.txt) - the option is per translation unit.
// LUM_SHADOW_KERNEL_EXTERN=1 (a flavour's main unit: wavefront_fast.hip, core.hip): the kernel is declared here and defined in wavefront_<flavour>_shadow.hip; the
// host stub and the code object come from the defining unit, the launch and hipFuncSetAttribute in wavefront_table_impl.h go through the declaration. The
// scheduler reorders instructions and changes none: the exact flavour's kernel stays bit-identical to the oracle.
#pragma once

#include "dev_trace.h"
#if LUM_PHASE_QUEUES
#include "dev_trace_pool.h"
#endif

#ifndef LUM_SHADOW_KERNEL_EXTERN
#define LUM_SHADOW_KERNEL_EXTERN 0
#endif

LUM_NS_BEGIN

struct ShadowQuery : ShadowState {
  ShadowQueue sq;
  const uint32_t* order;
  uint32_t out;
  uint32_t item;  // LUM_PHASE_QUEUES: the queue entry load() read
#if LUM_FAST
  static constexpr uint32_t kMutableVecs = 1;
  LUM_DEV void save_mutable(uint4* m) const { m[0] = make_uint4(fbits(tr), fbits(tg), fbits(tb), out); }
  LUM_DEV void load_mutable(const uint4* m, float tmax) { tr = bitsf(m[0].x); tg = bitsf(m[0].y); tb = bitsf(m[0].z); out = m[0].w; dist = tmax; blocked = false; }
#else
  static constexpr uint32_t kMutableVecs = 2;
  LUM_DEV void save_mutable(uint4* m) const {
    const unsigned long long r = (unsigned long long) __double_as_longlong(tr), g = (unsigned long long) __double_as_longlong(tg), b = (unsigned long long) __double_as_longlong(tb);
    m[0] = make_uint4((uint32_t) r, (uint32_t) (r >> 32), (uint32_t) g, (uint32_t) (g >> 32));
    m[1] = make_uint4((uint32_t) b, (uint32_t) (b >> 32), out, 0u);
  }
  LUM_DEV void load_mutable(const uint4* m, float tmax) {
    tr = __longlong_as_double((long long) ((unsigned long long) m[0].x | ((unsigned long long) m[0].y << 32)));
    tg = __longlong_as_double((long long) ((unsigned long long) m[0].z | ((unsigned long long) m[0].w << 32)));
    tb = __longlong_as_double((long long) ((unsigned long long) m[1].x | ((unsigned long long) m[1].y << 32)));
    out = m[1].z; dist = tmax; blocked = false;
  }
#endif
  LUM_DEV void save_const(uint4& c) const { c = make_uint4(tgt_inst, tgt_tri, self_inst, self_tri); }
  LUM_DEV void load_const(uint4 c) { tgt_inst = c.x; tgt_tri = c.y; self_inst = c.z; self_tri = c.w; }
  LUM_DEV void world_ray(const DeviceScene&, uint32_t j, V3& o, V3& d) const {
    const float4 o4 = sq.origin_dist[j], d4 = sq.dir_out[j];
    o = v3(o4.x, o4.y, o4.z); d = v3(d4.x, d4.y, d4.z);
  }
  LUM_DEV bool load(const DeviceScene&, uint32_t slot, V3& o, V3& d, float& tmax) {
    const uint32_t j = order ? order[slot] : slot;
    item = j;
    const float4 o4 = ld_stream(&sq.origin_dist[j]), d4 = ld_stream(&sq.dir_out[j]);
    begin(ld_stream(&sq.ids[j]), o4.w);
    out = fbits(d4.w);
    o = v3(o4.x, o4.y, o4.z); d = v3(d4.x, d4.y, d4.z); tmax = o4.w;
    return true;
  }
  LUM_DEV void finish(const DeviceScene&, uint32_t j) {
    const Col v = result();
#ifdef LUM_PHASE_STATS
    { const uint32_t kind = min(out / sq.capacity, 3u); atomicAdd(&g_vis_stat[2u * kind], 1ull); if (blocked) atomicAdd(&g_vis_stat[2u * kind + 1u], 1ull); }
#endif
#ifdef LUM_EXPERIMENT_VIS_IN_ITEM_ORDER
    sq.vis[j] = make_float4(v.r, v.g, v.b, 0.0f);
#else
    st_stream(&sq.vis[out], make_float4(v.r, v.g, v.b, 0.0f));
#endif
  }
};

#if LUM_SHADOW_KERNEL_EXTERN
__global__ LUM_TRACE_BOUNDS void k_shadow_rays(DeviceScene sc, ShadowQueue sq, const uint32_t* order, uint32_t* ctrl, uint64_t* counters, uint32_t lds_nodes);
#else
__global__ LUM_TRACE_BOUNDS void k_shadow_rays(DeviceScene sc, ShadowQueue sq, const uint32_t* order, uint32_t* ctrl, uint64_t* counters, uint32_t lds_nodes) {
  RayStats st{0, 0, 0};
  uint32_t rays = 0;
  ShadowQuery q;
  q.sq = sq;
  q.order = order;
#if LUM_PHASE_QUEUES
  trace_items_pool(sc, ctrl[kCtlShadowItems], ctrl + kCtlShadowCursor, q, st, rays, lds_nodes);
#else
  LUM_TRACE_ITEMS(sc, ctrl[kCtlShadowItems], ctrl + kCtlShadowCursor, q, st, rays, lds_nodes);
#endif
  flush_stats(counters, st, rays, kCntShadow, kCntNodesShadow, kCntTrisShadow, kCntNodesLdsShadow);
}
#endif

LUM_NS_END
