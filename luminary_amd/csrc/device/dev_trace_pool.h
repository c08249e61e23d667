// LUM_PHASE_QUEUES: the persistent two-level traversal with rays that are NOT bound to a lane.
//
// trace_items (dev_trace.h) keeps one ray per lane for the ray's life and lets a wave run the phase most of its lanes wait for; the others idle: VALU lane
// utilisation 0.45 (closest hits) / 0.53 (visibility) on the hall. Here every WAVE owns a pool of LUM_POOL_SLOTS rays (more rays than lanes) whose traversal
// state lives in LDS, and three lists of slot numbers - rays that wait for a node visit, for the triangles of a leaf, for an instance entry - plus the free
// slots. A wave iteration takes up to 64 slots from ONE list, loads their state, runs the unchanged phase code of dev_trace.h (visit_node, Q::on_tris, the pop
// loop) on a full wave and appends every slot to the list of its next phase. The pools are private to a wave: LDS operations of one wave complete in order, so
// the lists need no atomics, no barriers and no fences; list lengths live in scalar registers.
//
// Where the state is:
//   LDS, per slot, three 16-byte words   [inv.xyz, tmax] [origin.xyz, cur] [sp | in-instance << 15, stack top (1-2 words), item]      (every phase)
//   LDS, per slot                        the oldest LUM_POOL_STACK_ENTRIES 8-byte entries of its traversal stack (twice as many 4-byte ones)
//   global (sc.pool_state), per slot     [direction.xyz, instance] and the query's own words (nearest hit so far / transparency product, ignore handles):
//                                        read by triangle and entry phases only; 64 bytes per slot, 128 KB per workgroup - L2-resident
//   global (sc.pool_stack), per slot     the stack entries beyond the LDS ones
// The world-space ray is not kept: leaving an instance re-reads it from the queue the ray came from (Q::world_ray) - and only if the traversal goes on.
// Results do not depend on the order in which rays, nodes and leaves are processed (dev_trace.h header), and the arithmetic of a phase is the same code on the
// same values, so the exact flavour stays bit-identical to the oracle.
#pragma once

#include "dev_trace.h"

LUM_NS_BEGIN

// LUM_POOL_SLOTS (rays per wave, <= 256) and LUM_POOL_STACK_ENTRIES (8-byte stack entries per slot in LDS): dev_scene.h, which sizes the kernels' LDS by them
#ifndef LUM_POOL_REFILL
#define LUM_POOL_REFILL 32  // take new rays as soon as this many slots are free (and always when no list can fill a wave)
#endif
constexpr uint32_t kPoolSlots = LUM_POOL_SLOTS;
constexpr uint32_t kPoolWaves = (uint32_t) kTraceBlock / 64u;
constexpr uint32_t kPoolSlotsWg = kPoolSlots * kPoolWaves;
constexpr uint32_t kPoolStackBytes = LUM_POOL_STACK_ENTRIES * 8u * kPoolSlotsWg;
constexpr uint32_t kPoolStateBytes = 3u * 16u * kPoolSlotsWg;
constexpr uint32_t kPoolListBytes = 4u * kPoolSlots * kPoolWaves;
static_assert(kPoolSlots <= 256u && kPoolSlots >= 64u && (kPoolSlots & 3u) == 0u, "slot numbers are bytes; a wave must be able to fill itself");
static_assert(kPoolStackBytes + kPoolStateBytes + kPoolListBytes == LUM_LDS_STACK_BYTES, "dev_scene.h sizes the ray kernels' LDS beyond the tree top by this");
constexpr uint32_t kPoolStateVecs = 4u;  // uint4 per slot in sc.pool_state: 0 [d, inst], 1-2 the query's mutable words, 3 its constant ones

template <typename E> struct PoolStack {
  typedef typename StackWord<E>::W W;
  typedef __attribute__((address_space(3))) W* LdsPtr;
  LdsPtr lds;  // entry 0 of this slot; entries are kPoolSlotsWg words apart (a wave reads one entry of 64 slots: consecutive slots, consecutive banks)
  W* mem;      // the same beyond the LDS entries, in the workgroup's part of sc.pool_stack
  LUM_DEV void store(int i, E e) {
    constexpr int kLds = (int) (LUM_POOL_STACK_ENTRIES * 8u / sizeof(W));
    if (i < kLds) lds[(uint32_t) i * kPoolSlotsWg] = StackWord<E>::pack(e);
    else mem[(size_t) (uint32_t) (i - kLds) * kPoolSlotsWg] = StackWord<E>::pack(e);
  }
  LUM_DEV E load(int i) const {
    constexpr int kLds = (int) (LUM_POOL_STACK_ENTRIES * 8u / sizeof(W));
    if (i < kLds) return StackWord<E>::unpack(lds[(uint32_t) i * kPoolSlotsWg]);
    return StackWord<E>::unpack(mem[(size_t) (uint32_t) (i - kLds) * kPoolSlotsWg]);
  }
};

LUM_DEV uint32_t top_x(uint2 e) { return e.x; }
LUM_DEV uint32_t top_y(uint2 e) { return e.y; }
LUM_DEV uint32_t top_x(uint32_t e) { return e; }
LUM_DEV uint32_t top_y(uint32_t) { return 0u; }
LUM_DEV void top_set(uint2& e, uint32_t x, uint32_t y) { e = make_uint2(x, y); }
LUM_DEV void top_set(uint32_t& e, uint32_t x, uint32_t) { e = x; }

// A query type Q provides, besides load / on_tris / finish of dev_trace.h:
//   uint32_t item                       the queue entry of the ray load() read (kept with the slot; finish() and world_ray() go by it)
//   void world_ray(sc, item, o, d)      the ray again, from the queue
//   void save_mutable(uint4&), load_mutable(uint4, tmax)   what on_tris changes and finish reads
//   void save_const(uint4&), load_const(uint4)            what on_tris reads and nobody changes
template <class Q>
LUM_DEV void trace_items_pool(const DeviceScene& sc, uint32_t n, uint32_t* __restrict__ cursor, Q& q, RayStats& st, uint32_t& rays, uint32_t lds_count) {
  if (n == 0u) return;
  using SE = StackEntry<Q::kCull>;
  using E = typename SE::E;
  typedef typename StackWord<E>::W StackW;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  extern __shared__ float4 lds_top[];
  {
    const float4* __restrict__ g = reinterpret_cast<const float4*>(sc.bvh_nodes);
    for (uint32_t i = threadIdx.x; i < lds_count * (kNodeBytes / 16u); i += blockDim.x) lds_top[lds_slot_swizzle(i)] = g[i];
  }
  __shared__ float4 lds_leaves[4u * LUM_LDS_INSTANCES];
  const uint32_t staged_leaves = min(sc.tlas_num_leaves, (uint32_t) LUM_LDS_INSTANCES);
  for (uint32_t i = threadIdx.x; i < 4u * staged_leaves; i += blockDim.x) lds_leaves[i] = sc.tlas_leaves[i];
  char* const pool_lds = reinterpret_cast<char*>(lds_top) + (size_t) lds_count * kNodeBytes;
  typedef __attribute__((address_space(3))) uint8_t* LdsBytes;
  typedef float PoolVec __attribute__((ext_vector_type(4)));  // (the HIP vector classes cannot be assigned through an address-space-qualified pointer)
  typedef __attribute__((address_space(3))) PoolVec* LdsVec;
  typedef __attribute__((address_space(3))) uint32_t* LdsWord;
  const LdsVec state_v = (LdsVec) reinterpret_cast<PoolVec*>(pool_lds + kPoolStackBytes);                    // field f of slot s at [f * kPoolSlotsWg + s]
  struct StateRef {  // state[i] = float4 / float4 x = state[i]
    LdsVec p;
    struct Cell {
      LdsVec q;
      LUM_DEV void operator=(float4 v) { PoolVec t = {v.x, v.y, v.z, v.w}; *q = t; }
      LUM_DEV operator float4() const { const PoolVec t = *q; return make_float4(t.x, t.y, t.z, t.w); }
    };
    LUM_DEV Cell operator[](uint32_t i) const { return Cell{p + i}; }
  };
  const StateRef state{state_v};
  const LdsBytes lists = (LdsBytes) reinterpret_cast<uint8_t*>(pool_lds + kPoolStackBytes + kPoolStateBytes) + wave * 4u * kPoolSlots;  // this wave's four lists
  enum { kNode = 0, kTris = 1, kEnter = 2, kFree = 3 };
  for (uint32_t i = lane; i < kPoolSlots; i += 64u) lists[kFree * kPoolSlots + i] = (uint8_t) i;
  __syncthreads();
  uint32_t count[4] = {0u, 0u, 0u, kPoolSlots};  // wave-uniform
  const NodeSource nodes{sc.bvh_nodes, reinterpret_cast<const char*>(lds_top), lds_count};
  const uint32_t slot0 = wave * kPoolSlots;  // this wave's first slot in the workgroup
  uint4* const gstate = sc.pool_state + (size_t) blockIdx.x * kPoolSlotsWg * kPoolStateVecs;  // vec v of slot s at gstate[v * kPoolSlotsWg + s]
  StackW* const gstack = reinterpret_cast<StackW*>(sc.pool_stack) + (size_t) blockIdx.x * kPoolSlotsWg * kStackSize;

  const uint32_t waves = gridDim.x * kPoolWaves;
  uint32_t chunk = n / (waves * 2u);
  chunk = (min(max(chunk, 64u), LUM_CHUNK_MAX) + 63u) & ~63u;
  uint32_t chunk_next = 0, chunk_end = 0;
  bool more = true;

  // appends the slots of the lanes in `mask` to list `l`
  auto push = [&](int l, unsigned long long mask, bool mine, uint32_t s) {
    if (mine) lists[(uint32_t) l * kPoolSlots + count[l] + (uint32_t) __popcll(mask & below)] = (uint8_t) s;
    count[l] += (uint32_t) __popcll(mask);
  };
  // the pop loop of dev_trace.h on the loaded state; `left`: an instance was left on the way
  auto pop = [&](PoolStack<E>& stk, int& sp, E& top, float tmax, uint32_t& cur, bool& left) {
    bool again;
    E e;
    do {
      e = top;
      const bool done = SE::node(e) == kTraversalDone, leave = SE::node(e) == kLeaveInstance;
      if (!done) { sp--; top = stk.load(sp); }
      left |= leave;
      again = !done && (leave || !SE::reachable(e, tmax));
    } while (again);
    cur = SE::node(e);
  };
  // where a slot goes next, and the results of the rays that ended
  auto dispatch = [&](bool active, uint32_t s, uint32_t cur, bool in_inst) {
    const bool done = active && cur == kTraversalDone;
    const bool leaf = active && !done && (cur & kBvhLeafBit);
    const unsigned long long m_done = __ballot(done), m_tris = __ballot(leaf && in_inst), m_enter = __ballot(leaf && !in_inst), m_node = __ballot(active && !done && !leaf);
    push(kNode, m_node, active && !done && !leaf, s);
    push(kTris, m_tris, leaf && in_inst, s);
    push(kEnter, m_enter, leaf && !in_inst, s);
    push(kFree, m_done, done, s);
  };

  while (true) {
    const uint32_t waiting = count[kNode] + count[kTris] + count[kEnter];
    const bool can_fill = count[kNode] >= 64u || count[kTris] >= 64u || count[kEnter] >= 64u;
    // ---- refill: new rays into free slots ----
    if (more && count[kFree] > 0u && (count[kFree] >= LUM_POOL_REFILL || !can_fill)) {
      if (chunk_next >= chunk_end) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cursor, chunk);
        base = __builtin_amdgcn_readfirstlane(base);
        chunk_next = base;
        chunk_end = min(base + chunk, n);
        more = base < n;
      }
      if (more) {
        const uint32_t take = min(min(chunk_end - chunk_next, count[kFree]), 64u);
        const bool mine = lane < take;
        uint32_t s = 0;
        bool ok = false;
        if (mine) {
          s = lists[kFree * kPoolSlots + count[kFree] - take + lane];
          V3 wo, wd;
          float tmax;
          if (q.load(sc, chunk_next + lane, wo, wd, tmax)) {
            rays++;
            const uint32_t e = 0x7F800000u;
            const bool finite = (fbits(wo.x) & e) != e && (fbits(wo.y) & e) != e && (fbits(wo.z) & e) != e && (fbits(wd.x) & e) != e && (fbits(wd.y) & e) != e &&
                                (fbits(wd.z) & e) != e;
            if (finite) {
              TRay r;
              r.set(wo, wd);
              const uint32_t g = slot0 + s;
              state[g] = make_float4(r.inv.x, r.inv.y, r.inv.z, tmax);
              state[kPoolSlotsWg + g] = make_float4(wo.x, wo.y, wo.z, bitsf(0u));
              E top = SE::make(kTraversalDone, 0.0f);
              state[2u * kPoolSlotsWg + g] = make_float4(bitsf(0u), bitsf(top_x(top)), bitsf(top_y(top)), bitsf(q.item));
              gstate[g] = make_uint4(fbits(wd.x), fbits(wd.y), fbits(wd.z), kNoInstance);
              uint4 m[2], c;
              q.save_mutable(m);
              q.save_const(c);
              for (uint32_t v = 0; v < Q::kMutableVecs; v++) gstate[(1u + v) * kPoolSlotsWg + g] = m[v];
              gstate[3u * kPoolSlotsWg + g] = c;
              ok = true;
            }
            else q.finish(sc, chunk_next + lane);
          }
        }
        chunk_next += take;
        count[kFree] -= take;
        const unsigned long long m_ok = __ballot(ok), m_back = __ballot(mine && !ok);
        push(kNode, m_ok, ok, s);
        push(kFree, m_back, mine && !ok, s);
      }
      continue;
    }
    if (waiting == 0u) break;  // nothing in flight and nothing left to take

    // ---- phase choice: a list that fills the wave, triangles first (the phase with the most loads per lane); otherwise the longest list ----
    bool run_tris, run_node;
    if (count[kTris] >= 64u) { run_tris = true; run_node = false; }
    else if (count[kNode] >= 64u) { run_tris = false; run_node = true; }
    else if (count[kEnter] >= 64u) { run_tris = false; run_node = false; }
    else {
      run_tris = count[kTris] >= count[kNode] && count[kTris] >= count[kEnter];
      run_node = !run_tris && count[kNode] >= count[kEnter];
    }
    uint32_t take, first;  // (constant list indices: the lengths stay in scalar registers)
    if (run_tris) { take = min(count[kTris], 64u); count[kTris] -= take; first = kTris * kPoolSlots + count[kTris]; }
    else if (run_node) { take = min(count[kNode], 64u); count[kNode] -= take; first = kNode * kPoolSlots + count[kNode]; }
    else { take = min(count[kEnter], 64u); count[kEnter] -= take; first = kEnter * kPoolSlots + count[kEnter]; }
    const bool active = lane < take;
    uint32_t s = 0, g = slot0;
    if (active) { s = lists[first + lane]; g = slot0 + s; }
    PoolStack<E> stk{(typename PoolStack<E>::LdsPtr) reinterpret_cast<StackW*>(pool_lds) + g, gstack + g};
    uint32_t cur = kTraversalDone;
    bool in_inst = false;

    if (run_node) {
      if (active) {
        const float4 f0 = state[g], f1 = state[kPoolSlotsWg + g], f2 = state[2u * kPoolSlotsWg + g];
        TRay r;
        r.inv = v3(f0.x, f0.y, f0.z);
        r.o = v3(f1.x, f1.y, f1.z);
        r.noi = v3(-(r.o.x * r.inv.x), -(r.o.y * r.inv.y), -(r.o.z * r.inv.z));
        r.nx = (r.inv.x < 0.0f) ? 48u : 0u;  r.fx = 48u - r.nx;
        r.ny = (r.inv.y < 0.0f) ? 64u : 16u; r.fy = 80u - r.ny;
        r.nz = (r.inv.z < 0.0f) ? 80u : 32u; r.fz = 112u - r.nz;
        const float tmax = f0.w;
        cur = fbits(f1.w);
        const uint32_t spw = fbits(f2.x);
        int sp = (int) (spw & 0x7FFFu);
        in_inst = (spw & 0x8000u) != 0u;
        E top;
        top_set(top, fbits(f2.y), fbits(f2.z));
        st.nodes++;
        cur = visit_node<Q::kOrdered, Q::kCull, Q::kFarFirst>(nodes, cur, r, tmax, stk, sp, top, st);
        if (cur == kBvhEmpty) {
          bool left = false;
          pop(stk, sp, top, tmax, cur, left);
          if (left) {
            in_inst = false;
            if (cur != kTraversalDone) {  // back in world space with something left to visit there
              V3 wo, wd;
              q.world_ray(sc, fbits(f2.w), wo, wd);
              TRay w;
              w.set(wo, wd);
              state[g] = make_float4(w.inv.x, w.inv.y, w.inv.z, tmax);
              state[kPoolSlotsWg + g] = make_float4(wo.x, wo.y, wo.z, bitsf(cur));
              gstate[g] = make_uint4(fbits(wd.x), fbits(wd.y), fbits(wd.z), kNoInstance);
            }
          }
        }
        ((LdsWord) reinterpret_cast<uint32_t*>(pool_lds + kPoolStackBytes))[(kPoolSlotsWg + g) * 4u + 3u] = cur;
        state[2u * kPoolSlotsWg + g] = make_float4(bitsf((uint32_t) sp | (in_inst ? 0x8000u : 0u)), bitsf(top_x(top)), bitsf(top_y(top)), f2.w);
        if (cur == kTraversalDone) {
          q.item = fbits(f2.w);
          uint4 m[2];
          for (uint32_t v = 0; v < Q::kMutableVecs; v++) m[v] = gstate[(1u + v) * kPoolSlotsWg + g];
          q.load_mutable(m, tmax);
          q.finish(sc, q.item);
        }
      }
    }
    else if (run_tris) {
      if (active) {
        const float4 f0 = state[g], f1 = state[kPoolSlotsWg + g], f2 = state[2u * kPoolSlotsWg + g];
        const uint4 g0 = gstate[g];
        {
          uint4 m[2];
          for (uint32_t v = 0; v < Q::kMutableVecs; v++) m[v] = gstate[(1u + v) * kPoolSlotsWg + g];
          q.load_mutable(m, f0.w);
          q.load_const(gstate[3u * kPoolSlotsWg + g]);
        }
        q.item = fbits(f2.w);
        float tmax = f0.w;
        cur = fbits(f1.w);
        const uint32_t spw = fbits(f2.x);
        int sp = (int) (spw & 0x7FFFu);
        in_inst = true;
        E top;
        top_set(top, fbits(f2.y), fbits(f2.z));
        const V3 o = v3(f1.x, f1.y, f1.z), d = v3(bitsf(g0.x), bitsf(g0.y), bitsf(g0.z));
        if (q.on_tris(sc, g0.w, cur & 0x0FFFFFFFu, ((cur >> 28) & 0x7u) + 1u, o, d, tmax, st)) cur = kTraversalDone;
        else {
          bool left = false;
          pop(stk, sp, top, tmax, cur, left);
          if (left) {
            in_inst = false;
            if (cur != kTraversalDone) {
              V3 wo, wd;
              q.world_ray(sc, q.item, wo, wd);
              TRay w;
              w.set(wo, wd);
              state[g] = make_float4(w.inv.x, w.inv.y, w.inv.z, tmax);
              state[kPoolSlotsWg + g] = make_float4(wo.x, wo.y, wo.z, bitsf(cur));
              gstate[g] = make_uint4(fbits(wd.x), fbits(wd.y), fbits(wd.z), kNoInstance);
            }
          }
        }
        if (cur == kTraversalDone) q.finish(sc, q.item);
        else {
          uint4 m[2];
          q.save_mutable(m);
          for (uint32_t v = 0; v < Q::kMutableVecs; v++) gstate[(1u + v) * kPoolSlotsWg + g] = m[v];
          const LdsWord w = (LdsWord) reinterpret_cast<uint32_t*>(pool_lds + kPoolStackBytes);
          w[g * 4u + 3u] = fbits(tmax);
          w[(kPoolSlotsWg + g) * 4u + 3u] = cur;
          state[2u * kPoolSlotsWg + g] = make_float4(bitsf((uint32_t) sp | (in_inst ? 0x8000u : 0u)), bitsf(top_x(top)), bitsf(top_y(top)), f2.w);
        }
      }
    }
    else {  // kEnter: a top-level leaf = one instance: map the ray with its world->object matrix, push the way back
      if (active) {
        const float4 f0 = state[g], f1 = state[kPoolSlotsWg + g], f2 = state[2u * kPoolSlotsWg + g];
        const uint4 g0 = gstate[g];
        cur = fbits(f1.w);
        int sp = (int) (fbits(f2.x) & 0x7FFFu);
        E top;
        top_set(top, fbits(f2.y), fbits(f2.z));
        const uint32_t leaf_index = cur & 0x0FFFFFFFu;
        float4 r0, r1, r2, meta;
        if (leaf_index < staged_leaves) { const float4* leaf = lds_leaves + 4u * leaf_index; r0 = leaf[0]; r1 = leaf[1]; r2 = leaf[2]; meta = leaf[3]; }
        else { const float4* __restrict__ leaf = sc.tlas_leaves + 4u * leaf_index; r0 = leaf[0]; r1 = leaf[1]; r2 = leaf[2]; meta = leaf[3]; }
        const V3 wo = v3(f1.x, f1.y, f1.z), wd = v3(bitsf(g0.x), bitsf(g0.y), bitsf(g0.z));
        const float px = wo.x - r0.w, py = wo.y - r1.w, pz = wo.z - r2.w;
        const V3 oo = v3(mat_row_apply(r0.x, r0.y, r0.z, px, py, pz), mat_row_apply(r1.x, r1.y, r1.z, px, py, pz), mat_row_apply(r2.x, r2.y, r2.z, px, py, pz));
        const V3 od = v3(mat_row_apply(r0.x, r0.y, r0.z, wd.x, wd.y, wd.z), mat_row_apply(r1.x, r1.y, r1.z, wd.x, wd.y, wd.z),
                         mat_row_apply(r2.x, r2.y, r2.z, wd.x, wd.y, wd.z));
        TRay r;
        r.set(oo, od);
        stack_push(stk, sp, top, SE::make(kLeaveInstance, 0.0f));
        cur = fbits(meta.y);
        in_inst = true;
        state[g] = make_float4(r.inv.x, r.inv.y, r.inv.z, f0.w);
        state[kPoolSlotsWg + g] = make_float4(oo.x, oo.y, oo.z, bitsf(cur));
        state[2u * kPoolSlotsWg + g] = make_float4(bitsf((uint32_t) sp | 0x8000u), bitsf(top_x(top)), bitsf(top_y(top)), f2.w);
        gstate[g] = make_uint4(fbits(od.x), fbits(od.y), fbits(od.z), fbits(meta.x));
      }
    }
    dispatch(active, s, cur, in_inst);
  }
}

LUM_NS_END
