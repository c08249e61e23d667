// Offline measure of the host builder's trees: node visits, leaf visits and triangle tests per ray of a CPU walk that follows the device traversal
// (dev_trace.h: nearest child first, far children stacked, closest-hit rays cull by the hit distance, visibility rays stop at the first hit).
// The ray kernels' time follows the node visits (profiles/r03_ab_experiments.txt: +10 % visits = +28 % time on the hall), so builder changes can be judged
// here before they go to the GPU.
//   g++ -O2 -std=c++17 -fopenmp -I luminary_amd/csrc/host -o /tmp/bvh_quality tools/bvh_quality.cpp luminary_amd/csrc/host/bvh_build.cpp
//   /tmp/bvh_quality vertices.f32 [rays]       vertices.f32 = the mesh as the device scene holds it: 3 x float4 per triangle (tools/dump_mesh.py)
// Rays: origins on random triangles (pushed off the surface), cosine-distributed directions about the normal - what a path tracer's bounces look like.
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "bvh_build.h"

using namespace lum;

struct V { float x, y, z; };
static V sub(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V add(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V mul(V a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V norm(V a) { const float l = std::sqrt(dot(a, a)); return l > 0 ? mul(a, 1.0f / l) : V{0, 0, 1}; }

static float hit_tri(const float* p, V o, V d) {  // Moeller-Trumbore on p0, p1, p2 (float4 stride)
  const V p0{p[0], p[1], p[2]}, e1 = sub(V{p[4], p[5], p[6]}, p0), e2 = sub(V{p[8], p[9], p[10]}, p0);
  const V h = cross(d, e2);
  const float a = dot(e1, h);
  if (std::fabs(a) < 1e-20f) return INFINITY;
  const float f = 1.0f / a;
  const V s = sub(o, p0);
  const float u = f * dot(s, h);
  if (u < 0 || u > 1) return INFINITY;
  const V q = cross(s, e1);
  const float v = f * dot(d, q);
  if (v < 0 || u + v > 1) return INFINITY;
  const float t = f * dot(e2, q);
  return t > 1e-4f ? t : INFINITY;
}

struct Stats { double nodes = 0, leaves = 0, tris = 0, rays = 0, hits = 0; };

static void walk(const Bvh4& bvh, const std::vector<float>& verts, V o, V d, float tmax, bool any_hit, Stats& st) {
  const float inv[3] = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
  const float oo[3] = {o.x, o.y, o.z};
  struct E { uint32_t node; float t; };
  E stack[256];
  int sp = 0;
  uint32_t cur = 0;
  float best = tmax;
  st.rays++;
  while (true) {
    if (cur & kBvhLeafBit) {
      const uint32_t first = cur & 0x0FFFFFFFu, count = ((cur >> 28) & 7u) + 1u;
      st.leaves++;
      bool stop = false;
      for (uint32_t j = 0; j < count; j++) {
        st.tris++;
        const float t = hit_tri(&verts[(size_t) bvh.prims[first + j] * 12], o, d);
        if (t < best) { best = t; if (any_hit) { stop = true; break; } }
      }
      if (stop) { st.hits++; return; }
    }
    else {
      st.nodes++;
      const Bvh4Node& n = bvh.nodes[cur];
      float k[4]; uint32_t c[4];
      for (int j = 0; j < 4; j++) {
        c[j] = n.child[j];
        k[j] = INFINITY;
        if (c[j] == kBvhEmpty) continue;
        const float lo[3] = {n.lo_x[j], n.lo_y[j], n.lo_z[j]}, hi[3] = {n.hi_x[j], n.hi_y[j], n.hi_z[j]};
        float tn = 0.0f, tf = best;
        for (int a = 0; a < 3; a++) {
          float t0 = (lo[a] - oo[a]) * inv[a], t1 = (hi[a] - oo[a]) * inv[a];
          if (t0 > t1) std::swap(t0, t1);
          tn = std::max(tn, t0); tf = std::min(tf, t1);
        }
        if (tn <= tf) k[j] = tn;
      }
      for (int a = 0; a < 4; a++)  // sort by entry distance (4 entries)
        for (int b = a + 1; b < 4; b++) if (k[b] < k[a]) { std::swap(k[a], k[b]); std::swap(c[a], c[b]); }
      for (int j = 3; j >= 1; j--) if (k[j] < INFINITY) stack[sp++] = {c[j], k[j]};
      if (k[0] < INFINITY) { cur = c[0]; continue; }
    }
    bool found = false;
    while (sp > 0) { const E e = stack[--sp]; if (any_hit || e.t <= best) { cur = e.node; found = true; break; } }
    if (!found) break;
  }
  if (best < tmax) st.hits++;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: bvh_quality vertices.f32 [rays]\n"); return 1; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 1; }
  std::fseek(f, 0, SEEK_END);
  const size_t bytes = (size_t) std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<float> verts(bytes / 4);
  if (std::fread(verts.data(), 4, verts.size(), f) != verts.size()) return 1;
  std::fclose(f);
  const uint32_t nt = (uint32_t) (verts.size() / 12);
  const uint32_t nrays = argc > 2 ? (uint32_t) std::atoi(argv[2]) : 400000;
  std::vector<Aabb> boxes(nt);
  for (uint32_t t = 0; t < nt; t++) {
    Aabb b{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
    for (int v = 0; v < 3; v++) for (int a = 0; a < 3; a++) { const float x = verts[(size_t) t * 12 + 4 * v + a]; b.lo[a] = std::min(b.lo[a], x); b.hi[a] = std::max(b.hi[a], x); }
    boxes[t] = b;
  }
  const auto t0 = std::chrono::steady_clock::now();
#ifdef LUM_BVH_HAS_TRIANGLE_BUILD
  const Bvh4 bvh = build_bvh4_triangles(verts.data(), boxes.data(), nullptr, nt, kBvhLeafMaxTri, 26);
#else
  const Bvh4 bvh = build_bvh4(boxes.data(), nt, kBvhLeafMaxTri, 26);
#endif
  const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (bvh.nodes.empty()) { std::printf("build failed (too deep)\n"); return 1; }
  size_t kids = 0, leaf_refs = 0, leaf_count = 0;
  for (const Bvh4Node& n : bvh.nodes)
    for (int j = 0; j < 4; j++) if (n.child[j] != kBvhEmpty) { kids++; if (n.child[j] & kBvhLeafBit) { leaf_count++; leaf_refs += ((n.child[j] >> 28) & 7u) + 1u; } }
  std::printf("triangles %u  nodes %zu  children per node %.2f  leaves %zu  references %zu (%.3f per triangle)  depth %u  build %.2f s\n", nt, bvh.nodes.size(),
              (double) kids / bvh.nodes.size(), leaf_count, leaf_refs, (double) leaf_refs / nt, bvh.max_depth, build_s);
  Stats closest, shadow;
#pragma omp parallel
  {
    Stats c, s;
    std::mt19937 rng(1234u + 977u * (unsigned) (
#ifdef _OPENMP
        omp_get_thread_num()
#else
        0
#endif
        ));
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < (int64_t) nrays; i++) {
      const uint32_t t = (uint32_t) (U(rng) * nt) % nt;
      const float* p = &verts[(size_t) t * 12];
      const V p0{p[0], p[1], p[2]}, e1 = sub(V{p[4], p[5], p[6]}, p0), e2 = sub(V{p[8], p[9], p[10]}, p0);
      float u = U(rng), v = U(rng);
      if (u + v > 1) { u = 1 - u; v = 1 - v; }
      V n = norm(cross(e1, e2));
      if (U(rng) < 0.5f) n = mul(n, -1.0f);
      const V o = add(add(p0, add(mul(e1, u), mul(e2, v))), mul(n, 1e-3f));
      const float r1 = U(rng), r2 = U(rng), phi = 6.2831853f * r1, sr = std::sqrt(r2);
      const V tan = norm(std::fabs(n.x) < 0.9f ? cross(n, V{1, 0, 0}) : cross(n, V{0, 1, 0})), bit = cross(n, tan);
      const V d = norm(add(add(mul(tan, sr * std::cos(phi)), mul(bit, sr * std::sin(phi))), mul(n, std::sqrt(1 - r2))));
      walk(bvh, verts, o, d, INFINITY, false, c);
      walk(bvh, verts, o, d, INFINITY, true, s);
    }
#pragma omp critical
    {
      closest.nodes += c.nodes; closest.leaves += c.leaves; closest.tris += c.tris; closest.rays += c.rays; closest.hits += c.hits;
      shadow.nodes += s.nodes; shadow.leaves += s.leaves; shadow.tris += s.tris; shadow.rays += s.rays; shadow.hits += s.hits;
    }
  }
  std::printf("closest: nodes %.2f  leaves %.2f  triangles %.2f per ray (hit %.2f)\n", closest.nodes / closest.rays, closest.leaves / closest.rays, closest.tris / closest.rays, closest.hits / closest.rays);
  std::printf("any-hit: nodes %.2f  leaves %.2f  triangles %.2f per ray (hit %.2f)\n", shadow.nodes / shadow.rays, shadow.leaves / shadow.rays, shadow.tris / shadow.rays, shadow.hits / shadow.rays);
  return 0;
}
