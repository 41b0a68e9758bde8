// stl_order_map.hpp -- a uint64 -> small-record map whose ITERATION ORDER is that of libstdc++'s
// std::unordered_map<uint64_t, T> filled by the same sequence of emplace() calls.
//
// graph.bin's k-mer records are written in the iteration order of the reference's
// `unordered_map<uint64_t, kmerCovFreBitVec> mGraphKmerHashHapStrMap` (src/construct_index.cpp:880-900), so a byte-
// identical file needs that order.  std::unordered_map itself gives it, at ~450 ns per insert on a 3e7-key table (one
// malloc per node, dependent cache misses, a pointer-chasing walk of the whole list at every growth step).
//
// What fixes the order is libstdc++'s _Hashtable with unique keys, identity hash and no cached hash code
// (<bits/hashtable.h>: _M_insert_bucket_begin, _M_rehash_aux): all nodes sit on ONE singly linked list; a new node goes
// to the front of its bucket's run, or -- into an empty bucket -- to the front of the whole list; a rehash walks the
// list once and re-inserts every node by the same two rules.  Hence, for the sequence S of nodes a bucket array has
// seen (the list it was rehashed from, in list order, then the nodes inserted while it was in use, in insertion order):
//     the list = buckets in REVERSE order of their first appearance in S, each bucket's nodes in REVERSE order of S.
// So the order need not be maintained at all while inserting.  Here inserts go to an open-addressing index (records in
// insertion order); the growth steps are the library's own (std::__detail::_Prime_rehash_policy is asked at every
// insert, only its answers are recorded); order() replays them afterwards as one stable grouping pass per bucket
// array -- sequential reads, prefetched scatters, ~2x the final size in total.
// `tests/native/order_map_check.cpp` compares the result with std::unordered_map on random insert sequences.
#pragma once
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>

namespace vgh {

class StlOrderMap {
public:
    static constexpr uint32_t kNil = 0xFFFFFFFFu;

    explicit StlOrderMap(size_t payload_bytes) : payload_(payload_bytes), cells_(1u << 10, 0) {}

    size_t size() const { return keys_.size(); }

    // the record of `key` (created with a zeroed payload when absent) and whether it was created
    std::pair<uint32_t, bool> emplace(uint64_t key)
    {
        size_t s = slot(key);
        for (;; s = (s + 1) & (cells_.size() - 1)) {
            const uint32_t c = cells_[s];
            if (c == 0) break;
            if (keys_[c - 1] == key) return {c - 1, false};
        }
        // std::unordered_map asks its policy before linking the new node (_M_insert_unique_node)
        const auto rh = policy_._M_need_rehash(n_bkt_, keys_.size(), 1);
        if (rh.first) {
            n_bkt_ = rh.second;
            epochs_.push_back(Epoch{keys_.size(), n_bkt_});
        }
        const uint32_t id = (uint32_t)keys_.size();
        keys_.push_back(key);
        data_.resize(data_.size() + payload_, 0);
        cells_[s] = id + 1;
        if (2 * keys_.size() > cells_.size()) grow();
        return {id, true};
    }
    uint32_t find(uint64_t key) const   // kNil when absent
    {
        for (size_t s = slot(key);; s = (s + 1) & (cells_.size() - 1)) {
            const uint32_t c = cells_[s];
            if (c == 0) return kNil;
            if (keys_[c - 1] == key) return c - 1;
        }
    }
    void prefetch(uint64_t key) const { __builtin_prefetch(&cells_[slot(key)]); }
    void prefetch_record(uint32_t id) const
    {
        __builtin_prefetch(&keys_[id]);
        __builtin_prefetch(data_.data() + (size_t)id * payload_);
    }

    uint64_t key(uint32_t id) const { return keys_[id]; }
    uint8_t* payload(uint32_t id) { return data_.data() + (size_t)id * payload_; }
    const uint8_t* payload(uint32_t id) const { return data_.data() + (size_t)id * payload_; }

    // record ids in the order std::unordered_map would iterate
    std::vector<uint32_t> order() const
    {
        std::vector<uint32_t> list;   // the list the next bucket array is rehashed from
        struct Bucket { uint32_t count, start; };   // one cache line touch per record in either pass
        std::vector<uint32_t> seq, bkt, grouped;
        std::vector<Bucket> buckets;
        for (size_t e = 0; e < epochs_.size(); ++e) {
            const size_t from = epochs_[e].first_id;
            const size_t to = e + 1 < epochs_.size() ? epochs_[e + 1].first_id : keys_.size();
            const size_t n_bkt = epochs_[e].n_bkt;
            // S = old list, then the records inserted while this bucket array was in use
            seq.assign(list.begin(), list.end());
            for (size_t id = from; id < to; ++id) seq.push_back((uint32_t)id);
            const size_t m = seq.size();
            bkt.resize(m);
            buckets.assign(n_bkt, Bucket{0, kNil});
            // stable grouping by bucket, buckets in order of first appearance ...
            for (size_t i = 0; i < m; ++i) {
                if (i + 24 < m) __builtin_prefetch(&keys_[seq[i + 24]]);
                bkt[i] = (uint32_t)(keys_[seq[i]] % n_bkt);
            }
            for (size_t i = 0; i < m; ++i) {
                if (i + 16 < m) __builtin_prefetch(&buckets[bkt[i + 16]], 1);
                buckets[bkt[i]].count++;
            }
            grouped.resize(m);
            uint32_t running = 0;
            for (size_t i = 0; i < m; ++i) {
                if (i + 16 < m) __builtin_prefetch(&buckets[bkt[i + 16]], 1);
                Bucket& b = buckets[bkt[i]];
                if (b.start == kNil) {
                    b.start = running;
                    running += b.count;
                }
                grouped[b.start++] = seq[i];
            }
            // ... and the list is that sequence backwards
            list.resize(m);
            for (size_t i = 0; i < m; ++i) list[i] = grouped[m - 1 - i];
        }
        return list;
    }

private:
    struct Epoch { size_t first_id, n_bkt; };   // records first_id.. were linked into an array of n_bkt buckets

    size_t slot(uint64_t k) const { return (size_t)((k * 0x9E3779B97F4A7C15ULL) >> 24) & (cells_.size() - 1); }
    void grow()
    {
        std::vector<uint32_t> nc(cells_.size() * 4, 0);
        cells_.swap(nc);
        for (uint32_t id = 0; id < keys_.size(); ++id) {
            size_t s = slot(keys_[id]);
            while (cells_[s]) s = (s + 1) & (cells_.size() - 1);
            cells_[s] = id + 1;
        }
    }

    size_t payload_;
    std::__detail::_Prime_rehash_policy policy_;   // the library's own growth decisions (bucket counts, thresholds)
    size_t n_bkt_ = 1;                             // a default-constructed unordered_map has its single bucket
    std::vector<Epoch> epochs_;
    std::vector<uint32_t> cells_;                  // open addressing: record id + 1, 0 = empty
    std::vector<uint64_t> keys_;                   // records in insertion order
    std::vector<uint8_t> data_;
};

}  // namespace vgh
