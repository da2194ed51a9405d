// hash_order.hpp — iteration order of a libstdc++ std::unordered_map<int, T> as a function of the key insertion sequence.
//
// The reference emits the points of a cloud by iterating the hash map that holds the best point per cell
// (utils/pts_preprocess.h:85-89, :124-128), so the ORDER of a cloud is libstdc++'s node-list order, and the float
// sequential average of SC.cpp:60-64 depends on it.  The host pre-stage keeps the real container; the
// GPU pre-stage (prestage.hip) needs the same order without it.  libstdc++'s _Hashtable with unique keys, identity hash
// and no cached hash codes is a singly linked node list plus, per bucket, a pointer to the node BEFORE the bucket's
// first node (bits/hashtable.h: _M_insert_bucket_begin, _M_rehash_aux(unique)); both are restated here on index arrays.
// The growth schedule (the bucket count chosen when the element count crosses each threshold) is not restated: it is
// read off the real container of THIS libstdc++ by probe_bucket_schedule() — it depends on counts only, never on keys.
// tests/test_prestage.py pins the whole thing against std::unordered_map (through the oracle).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define PR_HD __host__ __device__
#else
#define PR_HD
#endif

namespace pr {

constexpr int HO_EMPTY = -2;   // bucket has no nodes
constexpr int HO_BB = -1;      // "before begin" sentinel / end of list

// keys[0..K): distinct keys in insertion order (non-negative).  sched_cnt[i] / sched_nb[i]: when a key is about to be
// inserted while sched_cnt[i] elements exist, the table is first rehashed to sched_nb[i] buckets (ascending, sched_cnt[0]
// == 0).  next[K], bkt[max bucket count] are scratch.  order[t] = index into keys of the t-th element of the iteration.
PR_HD inline void hash_order(const int* keys, int K, const int* sched_cnt, const int* sched_nb, int nsched, int* next,
                             int* bkt, int* order) {
  int head = HO_BB;   // before_begin.next
  int nb = 1, si = 0;
  for (int k = 0; k < K; k++) {
    if (si < nsched && sched_cnt[si] == k) {   // _M_rehash_aux(n, unique keys)
      nb = sched_nb[si++];
      for (int b = 0; b < nb; b++) bkt[b] = HO_EMPTY;
      int p = head, bbegin = 0;
      head = HO_BB;
      while (p != HO_BB) {
        const int nx = next[p];
        const int b = (int)((unsigned)keys[p] % (unsigned)nb);
        if (bkt[b] == HO_EMPTY) {
          next[p] = head;
          head = p;
          bkt[b] = HO_BB;
          if (next[p] != HO_BB) bkt[bbegin] = p;
          bbegin = b;
        } else {
          const int prev = bkt[b];
          if (prev == HO_BB) { next[p] = head; head = p; } else { next[p] = next[prev]; next[prev] = p; }
        }
        p = nx;
      }
    }
    const int b = (int)((unsigned)keys[k] % (unsigned)nb);   // _M_insert_bucket_begin
    if (bkt[b] != HO_EMPTY) {
      const int prev = bkt[b];
      if (prev == HO_BB) { next[k] = head; head = k; } else { next[k] = next[prev]; next[prev] = k; }
    } else {
      next[k] = head;
      head = k;
      if (next[k] != HO_BB) bkt[(int)((unsigned)keys[next[k]] % (unsigned)nb)] = k;
      bkt[b] = HO_BB;
    }
  }
  int t = 0;
  for (int p = head; p != HO_BB; p = next[p]) order[t++] = p;
}

}  // namespace pr
