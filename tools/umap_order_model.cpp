// umap_order_model.cpp — CPU check of the closed form behind ks_k_bundle_order.h.
//
// The reference integrates `merged` bundles in the iteration order of a libstdc++
// std::unordered_map (kimera_semantics/src/semantic_tsdf_integrator_merged.cpp:200-232).  That order is a
// pure function of (a) the sequence of distinct keys in first-insertion order, (b) their hash codes and
// (c) the rehash schedule (bucket counts b_e, taken over at element counts t_e):
//   epoch e re-inserts the list L_{e-1} (elements [0, t_e), in list order) and then the new elements
//   [t_e, t_{e+1}) into an empty table of b_e buckets; "insert" puts a node at the front of its bucket's
//   segment, and a bucket that becomes non-empty at the front of the whole list.  Hence
//     L_e = elements sorted by (first-occupation position of their bucket DESC, own position DESC),
//   where position = rank in L_{e-1} for re-inserted elements and the insertion index for new ones.
// This program draws random key sets, builds the real unordered_map, and compares its iteration order
// with the closed form evaluated epoch by epoch (the same recurrence the GPU kernels evaluate in parallel).
//   g++ -O2 -std=c++17 -o /tmp/umap_order_model tools/umap_order_model.cpp && /tmp/umap_order_model
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <unordered_map>
#include <vector>

struct Hash32 {  // stands for vxb::LongIndexHash: an arbitrary 32-bit code widened to size_t
  size_t operator()(uint64_t k) const { return (uint32_t)(k * 0x9E3779B97F4A7C15ull >> 29); }
};

// rehash schedule of the host's libstdc++, probed from a real container
static void probe_schedule(size_t n_max, std::vector<uint32_t>* t, std::vector<uint32_t>* b) {
  std::unordered_map<uint32_t, char> m;
  size_t bc = m.bucket_count();
  for (uint32_t i = 0; i < n_max; ++i) {
    m[i] = 0;
    if (m.bucket_count() != bc) {
      bc = m.bucket_count();
      t->push_back(i);  // the insertion of element index i triggered the rehash
      b->push_back((uint32_t)bc);
    }
  }
}

int main() {
  std::vector<uint32_t> T, Bk;
  probe_schedule(3000000, &T, &Bk);
  printf("schedule:");
  for (size_t e = 0; e < T.size(); ++e) printf(" t=%u->b=%u", T[e], Bk[e]);
  printf("\n");
  std::mt19937_64 rng(1);
  size_t checked = 0;
  for (int trial = 0; trial < 60; ++trial) {
    const size_t n = trial < 40 ? 1 + rng() % 3000 : 1 + rng() % 400000;
    const bool clustered = trial & 1;  // many equal hash codes modulo small primes
    std::vector<uint64_t> keys;
    std::unordered_map<uint64_t, int, Hash32> real;
    while (keys.size() < n) {
      uint64_t k = clustered ? (rng() % 64) * 7919u + (rng() % 5000) * 13 : rng();
      if (real.emplace(k, 0).second) keys.push_back(k);
    }
    std::vector<uint64_t> want;
    for (auto& kv : real) want.push_back(kv.first);
    // closed form
    const size_t B = keys.size();
    std::vector<uint32_t> H(B);
    for (size_t i = 0; i < B; ++i) H[i] = (uint32_t)Hash32()(keys[i]);
    std::vector<uint32_t> L;  // list order (element ids)
    for (size_t e = 0; e < T.size() && T[e] < B; ++e) {
      const size_t total = std::min<size_t>(e + 1 < T.size() ? T[e + 1] : ~0u, B);
      std::vector<uint32_t> pos(total);
      for (size_t r = 0; r < L.size(); ++r) pos[L[r]] = (uint32_t)r;
      for (size_t id = L.size(); id < total; ++id) pos[id] = (uint32_t)id;
      std::vector<uint32_t> fmin(Bk[e], ~0u);
      for (size_t id = 0; id < total; ++id) fmin[H[id] % Bk[e]] = std::min(fmin[H[id] % Bk[e]], pos[id]);
      std::vector<uint32_t> ids(total);
      for (size_t id = 0; id < total; ++id) ids[id] = (uint32_t)id;
      std::sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t c) {
        const uint32_t fa = fmin[H[a] % Bk[e]], fc = fmin[H[c] % Bk[e]];
        if (fa != fc) return fa > fc;
        return pos[a] > pos[c];
      });
      L.swap(ids);
    }
    if (L.size() != want.size()) { printf("size mismatch\n"); return 1; }
    for (size_t i = 0; i < B; ++i)
      if (keys[L[i]] != want[i]) { printf("MISMATCH trial %d n %zu at %zu\n", trial, n, i); return 1; }
    checked += B;
  }
  printf("OK: closed form == std::unordered_map iteration order on %zu keys\n", checked);
  return 0;
}
