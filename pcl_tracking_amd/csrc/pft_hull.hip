// pft_hull.hip -- host code: which reference points can attain a minimum or maximum of a rigidly transformed
// coordinate (A3, calcBoundingBox of /root/reference/src/auto_tracking.cpp's tracker: PCL transforms all M points of every
// particle's cloud and takes getMinMax3D over them).
//
// max over the cloud of r . m is attained at a vertex of the cloud's convex hull whatever the direction r, so the box
// of P particles needs the hull's vertices only: a few hundred of a 2 048-point scan instead of all of them.  The device
// evaluates fl((r0 x + r1 y) + r2 z) in float, so "vertex" is widened to a shell: a point is dropped only if it lies at
// least eps inside EVERY facet plane of the hull, eps = 8 x the rounding-error bound of that expression -- then its
// float value stays below the float value of the hull vertex that is extreme in the direction concerned, for every
// rotation row r (|r| <= 1 + 1e-6), and min / max over the subset equal min / max over the cloud bit for bit.
//
// The hull is built incrementally in double (every point against every LIVE facet: a reference cloud is a few thousand
// points, once per pft_set_reference) and then VERIFIED -- every point inside every facet, every edge shared by exactly
// two facets -- before anything is dropped; a cloud that is degenerate (planar, collinear, tiny, non-finite
// coordinates) or fails the verification keeps all its points.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "pft_internal.h"

namespace {
struct Face {
  uint32_t v[3];
  double n[3], d;  // unit outward normal, plane offset: n . x = d on the plane
  bool alive;
};

inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

bool make_face(const std::vector<double>& X, uint32_t a, uint32_t b, uint32_t c, const double* inside, Face& f,
               double tiny) {
  const double *A = &X[3 * a], *B = &X[3 * b], *C = &X[3 * c];
  const double u[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, w[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
  double n[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
  const double len = std::sqrt(dot3(n, n));
  if (!(len > tiny)) return false;  // a sliver: the caller gives up
  for (int k = 0; k < 3; k++) n[k] /= len;
  double d = dot3(n, A);
  f.v[0] = a;
  f.v[1] = b;
  f.v[2] = c;
  if (inside && dot3(n, inside) > d) {  // orient away from a point known to be inside
    std::swap(f.v[1], f.v[2]);
    for (int k = 0; k < 3; k++) n[k] = -n[k];
    d = -d;
  }
  for (int k = 0; k < 3; k++) f.n[k] = n[k];
  f.d = d;
  f.alive = true;
  return true;
}

inline uint64_t edge_key(uint32_t a, uint32_t b) { return ((uint64_t)a << 32) | b; }
}  // namespace

// indices (ascending) of the points kept for the box; all of them when nothing can be dropped safely
void pft_aabb_support_subset(const pft_point_xyzrgba* pts, size_t n, std::vector<uint32_t>& keep) {
  keep.clear();
  auto all = [&]() {
    keep.resize(n);
    for (size_t i = 0; i < n; i++) keep[i] = (uint32_t)i;
  };
  if (n < 128 || n > 8192) return all();  // nothing to gain / an O(n x facets) pass of seconds
  std::vector<double> X(3 * n);
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, l1max = 0.0;
  for (size_t i = 0; i < n; i++) {
    const float c[3] = {pts[i].x, pts[i].y, pts[i].z};
    for (int a = 0; a < 3; a++) {
      if (!std::isfinite(c[a])) return all();
      X[3 * i + a] = (double)c[a];
      lo[a] = std::min(lo[a], (double)c[a]);
      hi[a] = std::max(hi[a], (double)c[a]);
    }
    l1max = std::max(l1max, std::fabs((double)c[0]) + std::fabs((double)c[1]) + std::fabs((double)c[2]));
  }
  const double scale = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
  if (!(scale > 0.0)) return all();
  const double tol = 1.0e-11 * scale;  // coplanarity in double on float inputs
  // float evaluation of (r0 x + r1 y) + r2 z with |r| <= 1: three products and two sums, each rounded once
  const double err = 3.0 * std::ldexp(1.0, -23) * l1max;
  const double eps = 8.0 * err + 4.0 * tol;

  // ---- the first tetrahedron ----
  uint32_t i0 = 0, i1 = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (X[3 * i] < X[3 * i0]) i0 = i;
    if (X[3 * i] > X[3 * i1]) i1 = i;
  }
  if (i0 == i1) {  // no extent in x: take y
    for (uint32_t i = 0; i < n; i++) {
      if (X[3 * i + 1] < X[3 * i0 + 1]) i0 = i;
      if (X[3 * i + 1] > X[3 * i1 + 1]) i1 = i;
    }
  }
  if (i0 == i1) return all();
  const double* A = &X[3 * i0];
  double ab[3] = {X[3 * i1] - A[0], X[3 * i1 + 1] - A[1], X[3 * i1 + 2] - A[2]};
  const double abl = std::sqrt(dot3(ab, ab));
  if (!(abl > 1.0e-6 * scale)) return all();
  uint32_t i2 = i0;
  double best = 0.0;
  for (uint32_t i = 0; i < n; i++) {
    const double ap[3] = {X[3 * i] - A[0], X[3 * i + 1] - A[1], X[3 * i + 2] - A[2]};
    const double c[3] = {ab[1] * ap[2] - ab[2] * ap[1], ab[2] * ap[0] - ab[0] * ap[2], ab[0] * ap[1] - ab[1] * ap[0]};
    const double dist = std::sqrt(dot3(c, c)) / abl;
    if (dist > best) {
      best = dist;
      i2 = i;
    }
  }
  if (!(best > 1.0e-4 * scale)) return all();  // collinear
  Face base;
  if (!make_face(X, i0, i1, i2, nullptr, base, 1.0e-12 * scale * scale)) return all();
  uint32_t i3 = i0;
  best = 0.0;
  for (uint32_t i = 0; i < n; i++) {
    const double dist = std::fabs(dot3(base.n, &X[3 * i]) - base.d);
    if (dist > best) {
      best = dist;
      i3 = i;
    }
  }
  if (!(best > 1.0e-3 * scale)) return all();  // planar (or nearly): every point is on the hull anyway
  double cen[3];
  for (int k = 0; k < 3; k++) cen[k] = (X[3 * i0 + k] + X[3 * i1 + k] + X[3 * i2 + k] + X[3 * i3 + k]) / 4.0;

  std::vector<Face> F;
  std::unordered_map<uint64_t, uint32_t> edge;  // directed edge -> its alive face
  const double tiny = 1.0e-14 * scale * scale;
  auto add_face = [&](uint32_t a, uint32_t b, uint32_t c) -> bool {
    Face f;
    if (!make_face(X, a, b, c, cen, f, tiny)) return false;
    const uint32_t id = (uint32_t)F.size();
    F.push_back(f);
    for (int k = 0; k < 3; k++) edge[edge_key(f.v[k], f.v[(k + 1) % 3])] = id;
    return true;
  };
  if (!add_face(i0, i1, i2) || !add_face(i0, i1, i3) || !add_face(i0, i2, i3) || !add_face(i1, i2, i3)) return all();

  // ---- every other point: the facets it sees go, the horizon is joined to it ----
  std::vector<uint32_t> vis;
  std::vector<std::pair<uint32_t, uint32_t>> horizon;
  size_t n_alive = 4;
  for (uint32_t p = 0; p < n; p++) {
    if (p == i0 || p == i1 || p == i2 || p == i3) continue;
    vis.clear();
    for (uint32_t f = 0; f < F.size(); f++)
      if (F[f].alive && dot3(F[f].n, &X[3 * p]) - F[f].d > tol) vis.push_back(f);
    if (vis.empty()) continue;
    for (uint32_t f : vis) F[f].alive = false;
    horizon.clear();
    for (uint32_t f : vis)
      for (int k = 0; k < 3; k++) {
        const uint32_t a = F[f].v[k], b = F[f].v[(k + 1) % 3];
        const auto tw = edge.find(edge_key(b, a));
        if (tw == edge.end()) return all();  // (the surface was not closed: give up)
        if (F[tw->second].alive) horizon.push_back({a, b});
      }
    for (uint32_t f : vis)
      for (int k = 0; k < 3; k++) edge.erase(edge_key(F[f].v[k], F[f].v[(k + 1) % 3]));
    if (horizon.size() < 3) return all();
    for (const auto& e : horizon)
      if (!add_face(e.first, e.second, p)) return all();
    n_alive += horizon.size();
    n_alive -= vis.size();
    if (n_alive > n) return all();  // more than half of the points are hull vertices: nothing worth dropping
    // dead facets are dropped from the list once they outnumber the live ones, so that a point is tested against O(live
    // facets): a shell-like cloud of 8 192 points costs ~2e7 plane tests up to the bail-out above instead of ~1e8
    if (F.size() > 2 * n_alive + 256) {
      std::vector<uint32_t> remap(F.size(), 0xffffffffu);
      std::vector<Face> G;
      G.reserve(2 * n_alive + 512);
      for (uint32_t f = 0; f < F.size(); f++)
        if (F[f].alive) {
          remap[f] = (uint32_t)G.size();
          G.push_back(F[f]);
        }
      F.swap(G);
      for (auto& kv : edge) {  // (the edges of dead facets were erased above: every entry names a live one)
        if (remap[kv.second] == 0xffffffffu) return all();
        kv.second = remap[kv.second];
      }
    }
  }

  // ---- verification: a closed surface (every edge has its twin) that no point is outside of ----
  size_t alive = 0;
  for (const Face& f : F) {
    if (!f.alive) continue;
    alive++;
    for (int k = 0; k < 3; k++) {
      const auto me = edge.find(edge_key(f.v[k], f.v[(k + 1) % 3]));
      const auto tw = edge.find(edge_key(f.v[(k + 1) % 3], f.v[k]));
      if (me == edge.end() || tw == edge.end() || !F[tw->second].alive) return all();
    }
  }
  if (alive < 4 || edge.size() != 3 * alive) return all();
  std::vector<double> outer(n, -INFINITY);  // signed distance to the nearest facet plane from inside (<= 0 inside)
  for (const Face& f : F) {
    if (!f.alive) continue;
    for (size_t i = 0; i < n; i++) {
      const double s = dot3(f.n, &X[3 * i]) - f.d;
      if (s > outer[i]) outer[i] = s;
    }
  }
  for (size_t i = 0; i < n; i++)
    if (outer[i] > 2.0 * tol) return all();  // a point outside the surface: not the hull

  // ---- the shell ----
  for (size_t i = 0; i < n; i++)
    if (outer[i] >= -eps) keep.push_back((uint32_t)i);
  if (keep.size() < 4) return all();
}
