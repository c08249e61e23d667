// The persistent two-level traversal on 8-wide nodes with children in octant slots (Bvh8oNode, dev_scene.h; LUM_BVH8O). Same queries, same work
// distribution, same phase vote as trace_items (dev_trace.h); what differs is the node visit and what the stack holds.
//   Reference for the scheme: the reference's own software traversal (unused there: OptiX does the work) - node layout src/luminary/utils.h:123-138,
//   child order by `slot ^ octant` and the group entries cuda/bvh.cuh:82-106, :146 - after Ylitie, Karras, Laine (HPG 2017).
// A visit reads 5 x 16 bytes (4-wide float nodes: 7), tests eight quantised boxes and leaves TWO masks in priority order (bit r = slot r ^ oct: the lowest set
// bit is the child the ray meets first): the inner children that may be hit - the lane's current GROUP {child_base, mask | imask}, of which it takes the
// first and stacks the rest as ONE entry - and the leaf slots that may be hit, whose primitives the triangle (or instance-entry) phase takes one slot at
// a time before the lane walks on. No entry distances: nothing is sorted, and a group popped later is walked even if the hit found meanwhile lies in
// front of it (each of its nodes still tests its children against the current hit distance).
//   Lane state: g_base, g_bits (inner hits 0-7 | imask 8-15 | leaf hits 16-23 | tag 30-31), l_node (the node the pending leaf slots belong to), l_cur
//   (the next leaf as a leaf word: kBvhLeafBit | (count - 1) << 28 | first primitive; kNoLeaf = fetch it from l_node's link word first), the newest stack
//   entry in registers, older ones in LDS / scratch as in trace_items.
#pragma once

LUM_NS_BEGIN

constexpr uint32_t kTagShift = 30u, kTagGroup = 0u, kTagMarker = 1u, kTagDone = 3u;
constexpr uint32_t kNoLeaf = 0u;  // a leaf word always carries kBvhLeafBit

// bit s -> bit s ^ oct in both bytes of a 16-bit word (three conditional swaps of neighbouring bit groups)
LUM_DEV uint32_t to_priority_order(uint32_t w, uint32_t oct) {
  w = (oct & 1u) ? (((w & 0x5555u) << 1) | ((w >> 1) & 0x5555u)) : w;
  w = (oct & 2u) ? (((w & 0x3333u) << 2) | ((w >> 2) & 0x3333u)) : w;
  w = (oct & 4u) ? (((w & 0x0F0Fu) << 4) | ((w >> 4) & 0x0F0Fu)) : w;
  return w;
}

struct Node8Words { float4 head; uint4 link, q0, q1, q2; };
LUM_DEV Node8Words load_node8(const NodeSource& src, uint32_t id, RayStats& st) {
  Node8Words n;
  const uint32_t b = id << 7;
  if (id < src.lds_count) {
    const char* p = src.lds + b;
    n.head = *reinterpret_cast<const float4*>(p); n.link = *reinterpret_cast<const uint4*>(p + 16u);
    n.q0 = *reinterpret_cast<const uint4*>(p + 32u); n.q1 = *reinterpret_cast<const uint4*>(p + 48u); n.q2 = *reinterpret_cast<const uint4*>(p + 64u);
    st.lds_nodes++;
  }
  else {
    const char* __restrict__ p = reinterpret_cast<const char*>(src.global) + b;
    n.head = *reinterpret_cast<const float4*>(p); n.link = *reinterpret_cast<const uint4*>(p + 16u);
    n.q0 = *reinterpret_cast<const uint4*>(p + 32u); n.q1 = *reinterpret_cast<const uint4*>(p + 48u); n.q2 = *reinterpret_cast<const uint4*>(p + 64u);
  }
  return n;
}
LUM_DEV uint4 load_link8(const NodeSource& src, uint32_t id) {
  const uint32_t b = (id << 7) + 16u;
  if (id < src.lds_count) return *reinterpret_cast<const uint4*>(src.lds + b);
  return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src.global) + b);
}
// the leaf word of slot `slot` of a node: link = child_base | leaf_base | meta[0..3] | meta[4..7]
LUM_DEV uint32_t leaf_word8(uint4 link, uint32_t slot) {
  const uint32_t m = ((slot < 4u ? link.z : link.w) >> ((slot & 3u) * 8u)) & 0xFFu;
  return kBvhLeafBit | ((m >> 5) << 28) | (link.y + (m & 31u));
}

// Tests the eight children of `n`: hit mask in SLOT order (empty slots carry inverted boxes and fail on their own).
LUM_DEV uint32_t test_node8(const Node8Words& n, const TRay& r, float tmax) {
  const uint32_t ew = fbits(n.head.w);
  const float sx = bitsf((ew & 0xFFu) << 23) * r.inv.x, sy = bitsf(((ew >> 8) & 0xFFu) << 23) * r.inv.y, sz = bitsf(((ew >> 16) & 0xFFu) << 23) * r.inv.z;
  const float bx = __builtin_fmaf(n.head.x, r.inv.x, r.noi.x), by = __builtin_fmaf(n.head.y, r.inv.y, r.noi.y), bz = __builtin_fmaf(n.head.z, r.inv.z, r.noi.z);
  // near / far planes per axis by the direction's sign: lo_x = q0.xy, lo_y = q0.zw, lo_z = q1.xy, hi_x = q1.zw, hi_y = q2.xy, hi_z = q2.zw
  const bool nx = (r.oct & 1u) != 0, ny = (r.oct & 2u) != 0, nz = (r.oct & 4u) != 0;
  const uint32_t ax0 = nx ? n.q1.z : n.q0.x, ax1 = nx ? n.q1.w : n.q0.y, fx0 = nx ? n.q0.x : n.q1.z, fx1 = nx ? n.q0.y : n.q1.w;
  const uint32_t ay0 = ny ? n.q2.x : n.q0.z, ay1 = ny ? n.q2.y : n.q0.w, fy0 = ny ? n.q0.z : n.q2.x, fy1 = ny ? n.q0.w : n.q2.y;
  const uint32_t az0 = nz ? n.q2.z : n.q1.x, az1 = nz ? n.q2.w : n.q1.y, fz0 = nz ? n.q1.x : n.q2.z, fz1 = nz ? n.q1.y : n.q2.w;
  uint32_t hits = 0;
#pragma unroll
  for (uint32_t j = 0; j < 8; j++) {
    const uint32_t wnx = j < 4 ? ax0 : ax1, wny = j < 4 ? ay0 : ay1, wnz = j < 4 ? az0 : az1, wfx = j < 4 ? fx0 : fx1, wfy = j < 4 ? fy0 : fy1, wfz = j < 4 ? fz0 : fz1;
    const float tnx = __builtin_fmaf(byte_f(wnx, j & 3u), sx, bx), tny = __builtin_fmaf(byte_f(wny, j & 3u), sy, by), tnz = __builtin_fmaf(byte_f(wnz, j & 3u), sz, bz);
    const float tfx = __builtin_fmaf(byte_f(wfx, j & 3u), sx, bx), tfy = __builtin_fmaf(byte_f(wfy, j & 3u), sy, by), tfz = __builtin_fmaf(byte_f(wfz, j & 3u), sz, bz);
    const float tn = vmax3(tnx, tny, vmax0(tnz)), tf = vmin3(tfx, tfy, vmin2(tfz, tmax));
    hits |= (tn <= __builtin_fmaf(tf, 1.000004f, 1e-30f)) ? (1u << j) : 0u;
  }
  return hits;
}

template <class Q>
LUM_DEV void trace_items8(const DeviceScene& sc, uint32_t n, uint32_t* __restrict__ cursor, Q& q, RayStats& st, uint32_t& rays, uint32_t lds_count) {
  if (n == 0u) return;
  typedef uint2 E;
  E stack_in_scratch[kStackSize];
  int sp = 0;
  TRay r;
  V3 wo = v3(0.0f, 0.0f, 0.0f), wd = v3(0.0f, 0.0f, 1.0f);
  float tmax = 0.0f;
  uint32_t inst = kNoInstance, idx = 0;
  uint32_t g_base = 0, g_bits = kTagDone << kTagShift, l_node = 0, l_cur = kNoLeaf;
  r.set(wo, wd);
  bool more = true;
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  extern __shared__ float4 lds_top[];
  {
    const float4* __restrict__ g = reinterpret_cast<const float4*>(sc.bvh_nodes);
    for (uint32_t i = threadIdx.x; i < lds_count * (kNodeBytes / 16u); i += blockDim.x) lds_top[i] = g[i];
    __syncthreads();
  }
  __shared__ float4 lds_leaves[4u * LUM_LDS_INSTANCES];
  const uint32_t staged_leaves = min(sc.tlas_num_leaves, (uint32_t) LUM_LDS_INSTANCES);
  for (uint32_t i = threadIdx.x; i < 4u * staged_leaves; i += blockDim.x) lds_leaves[i] = sc.tlas_leaves[i];
  __syncthreads();
  const NodeSource nodes{sc.bvh_nodes, reinterpret_cast<const char*>(lds_top), lds_count};
  typedef typename TraversalStack<E>::W StackW;
  TraversalStack<E> stk{(typename TraversalStack<E>::ScratchPtr) reinterpret_cast<StackW*>(stack_in_scratch),
                        (typename TraversalStack<E>::LdsPtr) (reinterpret_cast<StackW*>(reinterpret_cast<char*>(lds_top) + lds_count * kNodeBytes) + threadIdx.x),
                        (int) (LUM_LDS_STACK_BYTES / (kRayBlockMax * (uint32_t) sizeof(E)))};
  E top = make_uint2(0u, kTagDone << kTagShift);
  auto is_done = [&]() { return (g_bits >> kTagShift) == kTagDone; };
  // starts the walk of a tree at node `root`: a group whose one inner child is that node (slot = oct, so that the rank among the inner slots is 0)
  auto start_at = [&](uint32_t root) { g_base = root; g_bits = 1u | ((1u << r.oct) << 8); l_cur = kNoLeaf; };

  const uint32_t waves = gridDim.x * (blockDim.x / 64u);
  uint32_t chunk = n / (waves * 2u);
  chunk = (min(max(chunk, 64u), LUM_CHUNK_MAX) + 63u) & ~63u;
  uint32_t chunk_next = 0, chunk_end = 0;

  while (true) {
    const unsigned long long idle = __ballot(is_done());
    if (idle != 0ull && more) {  // wave-uniform
      if (chunk_next >= chunk_end) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cursor, chunk);
        base = __builtin_amdgcn_readfirstlane(base);
        chunk_next = base;
        chunk_end = min(base + chunk, n);
        more = base < n;
      }
      if (more) {
        const uint32_t avail = chunk_end - chunk_next, want = (uint32_t) __popcll(idle);
        const uint32_t rank = (uint32_t) __popcll(idle & below);
        if (is_done() && rank < avail) {
          idx = chunk_next + rank;
          if (q.load(sc, idx, wo, wd, tmax)) {
            rays++;
            const uint32_t e = 0x7F800000u;
            const bool finite = (fbits(wo.x) & e) != e && (fbits(wo.y) & e) != e && (fbits(wo.z) & e) != e && (fbits(wd.x) & e) != e && (fbits(wd.y) & e) != e &&
                                (fbits(wd.z) & e) != e;
            if (finite) { r.set(wo, wd); sp = 0; top = make_uint2(0u, kTagDone << kTagShift); inst = kNoInstance; start_at(0u); }
            else q.finish(sc, idx);
          }
        }
        chunk_next += min(want, avail);
      }
    }
    if (__ballot(!is_done()) == 0ull) break;

    while (true) {
      const bool live = !is_done();
      const bool has_leaf = live && (l_cur != kNoLeaf || ((g_bits >> 16) & 0xFFu) != 0u);
      const bool want_tris = has_leaf && inst != kNoInstance, want_enter = has_leaf && inst == kNoInstance;
      const uint32_t n_live = (uint32_t) __popcll(__ballot(live)), n_tris = (uint32_t) __popcll(__ballot(want_tris)), n_enter = (uint32_t) __popcll(__ballot(want_enter));
      if (n_live == 0u) break;
      const uint32_t n_nodes = n_live - n_tris - n_enter;
      const bool run_tris = n_tris * LUM_VOTE_TRIS >= max(n_nodes, n_enter) * LUM_VOTE_NODES;
      const bool run_enter = !run_tris && n_enter >= n_nodes;
      const bool do_tris = run_tris && want_tris, do_enter = run_enter && want_enter, do_node = !run_tris && !run_enter && live && !has_leaf;
      if (do_tris || do_enter) {
        // the next leaf slot of l_node, nearest first (its primitives are consecutive from leaf_base + offset): the one the visit prepared, or fetched now
        if (l_cur == kNoLeaf) {
          const uint32_t lb = (g_bits >> 16) & 0xFFu;
          const uint32_t rb = (uint32_t) __builtin_ctz(lb);
          g_bits &= ~(1u << (16u + rb));
          l_cur = leaf_word8(load_link8(nodes, l_node), rb ^ r.oct);
        }
      }
      {
        if (do_tris) {
          const uint32_t leaf = l_cur;
          l_cur = kNoLeaf;
          if (q.on_tris(sc, inst, leaf & 0x0FFFFFFFu, ((leaf >> 28) & 0x7u) + 1u, r.o, r.d, tmax, st)) { g_bits = kTagDone << kTagShift; q.finish(sc, idx); }
        }
      }
      {
        if (do_enter) {
          const uint32_t leaf_index = l_cur & 0x0FFFFFFFu;
          l_cur = kNoLeaf;
          float4 r0, r1, r2, meta;
          if (leaf_index < staged_leaves) { const float4* leaf = lds_leaves + 4u * leaf_index; r0 = leaf[0]; r1 = leaf[1]; r2 = leaf[2]; meta = leaf[3]; }
          else { const float4* __restrict__ leaf = sc.tlas_leaves + 4u * leaf_index; r0 = leaf[0]; r1 = leaf[1]; r2 = leaf[2]; meta = leaf[3]; }
          // what is left of the top level waits below a marker: the group (inner and leaf slots still to be walked) and the node its leaf slots belong to
          stk.store(sp, top); sp++; top = make_uint2(g_base, g_bits);
          stk.store(sp, top); sp++; top = make_uint2(l_node, kTagMarker << kTagShift);
          inst = fbits(meta.x);
          const float px = wo.x - r0.w, py = wo.y - r1.w, pz = wo.z - r2.w;
          const V3 oo = v3(mat_row_apply(r0.x, r0.y, r0.z, px, py, pz), mat_row_apply(r1.x, r1.y, r1.z, px, py, pz), mat_row_apply(r2.x, r2.y, r2.z, px, py, pz));
          const V3 od = v3(mat_row_apply(r0.x, r0.y, r0.z, wd.x, wd.y, wd.z), mat_row_apply(r1.x, r1.y, r1.z, wd.x, wd.y, wd.z),
                           mat_row_apply(r2.x, r2.y, r2.z, wd.x, wd.y, wd.z));
          r.set(oo, od);
          start_at(fbits(meta.y));
        }
      }
      {
        if (do_node) {
          // the group at hand, or the newest one with something left (a marker on the way leads back to the top level)
          bool pending_leaf = false;
          while ((g_bits & 0xFFu) == 0u) {
            const uint2 e = top;
            const uint32_t tag = e.y >> kTagShift;
            if (tag == kTagDone) { g_bits = kTagDone << kTagShift; break; }
            sp--; top = stk.load(sp);
            if (tag == kTagMarker) { l_node = e.x; inst = kNoInstance; r.set(wo, wd); continue; }
            g_base = e.x; g_bits = e.y;
            if (((g_bits >> 16) & 0xFFu) != 0u) { pending_leaf = true; break; }  // top-level leaf slots (instances) come before the group's other nodes
          }
          if (is_done()) q.finish(sc, idx);
          else if (!pending_leaf) {
            const uint32_t inner = g_bits & 0xFFu;
            const uint32_t rb = Q::kFarFirst ? (31u - (uint32_t) __builtin_clz(inner)) : (uint32_t) __builtin_ctz(inner);
            g_bits &= ~(1u << rb);
            const uint32_t slot = rb ^ r.oct;
            const uint32_t child = g_base + (uint32_t) __builtin_popcount((g_bits >> 8) & 0xFFu & ((1u << slot) - 1u));
            if ((g_bits & 0xFFu) != 0u) { stk.store(sp, top); sp++; top = make_uint2(g_base, g_bits & 0xFFFFu); }  // the rest of the group: one entry
            st.nodes++;
            const Node8Words nw = load_node8(nodes, child, st);
            const uint32_t hits = test_node8(nw, r, tmax);
            const uint32_t imask = fbits(nw.head.w) >> 24;
            const uint32_t prio = to_priority_order((hits & imask) | ((hits & ~imask & 0xFFu) << 8), r.oct);
            g_base = nw.link.x;
            g_bits = (prio & 0xFFu) | (imask << 8) | ((prio >> 8) << 16);
            l_node = child;
            l_cur = kNoLeaf;
            const uint32_t lb = prio >> 8;
            if (lb != 0u) {  // the nearest leaf slot's word while the link word is at hand
              const uint32_t rl = (uint32_t) __builtin_ctz(lb);
              g_bits &= ~(1u << (16u + rl));
              l_cur = leaf_word8(nw.link, rl ^ r.oct);
            }
          }
        }
      }
      if (more && n_live < LUM_REFILL) break;
    }
  }
}

LUM_NS_END
