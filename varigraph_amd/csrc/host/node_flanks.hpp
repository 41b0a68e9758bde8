// node_flanks.hpp -- the sequence a haplotype carries immediately left and right of a node
// (construct_index::find_node_up_down_seq, src/construct_index.cpp:1266-1549): used by `construct` to index a node's
// k-mers per haplotype and by the genotyping HMM for the multi-copy k-mer check.  NodeT needs `.start` (1-based node
// start) and `.gn` (const GraphNode*: seqs, hap_gt); `nodes` = every node of the chromosome in start order.
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "graph_index.hpp"

namespace vgh {

// `alt_seq` may be patched: a single-base alternative allele inside a reference-allele node overrides that base.
template <typename NodeT>
std::pair<std::string, std::string> node_flanks(const std::vector<NodeT>& nodes, uint32_t node_i, uint16_t hap, uint16_t alt_gt,
                                                std::string& alt_seq, uint32_t want)
{
    const NodeT& self = nodes[node_i];
    const uint32_t alt_start = self.start;
    const uint32_t alt_end = (uint32_t)(alt_start + self.gn->seqs[0].size() - 1);
    const uint32_t alt_len = (uint32_t)alt_seq.size();
    std::string up, down;

    auto allele_of = [&](const NodeT& n) -> uint16_t {
        const uint16_t gt = hap < n.gn->hap_gt.size() ? n.gn->hap_gt[hap] : 0;
        if (gt >= n.gn->seqs.size())
            throw std::runtime_error("The node '" + std::to_string(alt_start) + "' lacks sequence information for haplotype " +
                                     std::to_string(gt) + ".");
        return gt;
    };

    // ---- upstream: walk left, newest piece goes to the front of `up`
    std::vector<uint32_t> piece_len = {alt_len};
    std::vector<uint16_t> piece_gt = {alt_gt};
    std::vector<uint32_t> piece_start = {alt_start};
    std::vector<uint32_t> piece_end = {alt_end};
    for (uint32_t i = node_i; up.size() < want && i > 0;) {
        --i;
        const NodeT& n = nodes[i];
        const uint32_t n_start = n.start;
        const uint32_t n_end = (uint32_t)(n_start + n.gn->seqs.at(0).size() - 1);
        const uint16_t gt = allele_of(n);
        std::string seq = n.gn->seqs[gt];
        while (piece_start.size() > 0 && n_end >= piece_start.back() && !seq.empty()) {
            if (gt == 0) {   // reference allele: cut it where the piece to its right begins
                seq = seq.substr(0, piece_start.back() - n_start);
                break;
            } else if (piece_gt.back() == 0 && !up.empty()) {
                // the piece to the right was taken as reference sequence but this node's alternative allele covers
                // part of it: drop the overlapped bases and reconsider
                const uint32_t drop = std::min(n_end - piece_start.back() + 1, piece_len.back());
                up = up.substr(drop, up.size() - drop);
                piece_len.pop_back();
                piece_gt.pop_back();
                piece_start.pop_back();
                piece_end.pop_back();
                continue;
            }
            break;
        }
        if (seq.empty()) continue;
        piece_start.push_back(n_start);
        piece_end.push_back(n_end);
        const int64_t remaining = (int64_t)want - (int64_t)up.size();
        if ((int64_t)seq.size() >= remaining) {
            up.insert(0, seq.substr(seq.size() - remaining, remaining));
            piece_len.push_back((uint32_t)remaining);
        } else {
            up.insert(0, seq);
            piece_len.push_back((uint32_t)seq.size());
        }
        piece_gt.push_back(gt);
    }

    // ---- downstream
    piece_len = {alt_len};
    piece_gt = {alt_gt};
    piece_start = {alt_start};
    piece_end = {alt_end};
    uint16_t prev_gt = alt_gt;
    for (uint32_t i = node_i; down.size() < want && ++i < nodes.size();) {
        const NodeT& n = nodes[i];
        const uint32_t n_start = n.start;
        const uint32_t n_len = (uint32_t)n.gn->seqs[0].size();
        const uint32_t n_end = n_start + n_len - 1;
        const uint16_t gt = allele_of(n);
        std::string seq = n.gn->seqs[gt];
        // a single-base alternative allele inside this (reference-allele) node replaces that base
        if (alt_gt == 0 && gt != 0 && n_end <= alt_end && seq.size() == 1 && n_len == 1)
            alt_seq.replace(n_start - alt_start, n_len, seq);
        if (n_end <= alt_end) continue;
        while (piece_end.size() > 0 && n_end <= piece_end.back() && !seq.empty()) {   // nested in the piece to its left
            if (gt == 0) {
                seq = "";
                break;
            } else if (prev_gt == 0 && !down.empty()) {
                const uint32_t drop = std::min(piece_end.back() - n_start + 1, piece_len.back());
                down = down.substr(0, down.size() - drop);
                piece_len.pop_back();
                piece_gt.pop_back();
                piece_start.pop_back();
                piece_end.pop_back();
                continue;
            }
            break;
        }
        while (piece_end.size() > 0 && n_start <= piece_end.back() && !seq.empty()) {   // overlaps the piece to its left
            if (gt == 0) {
                seq = seq.substr(piece_end.back() - n_start + 1, n_end - piece_end.back());
                break;
            } else if (prev_gt == 0 && !down.empty()) {
                const uint32_t drop = std::min(piece_end.back() - n_start + 1, piece_len.back());
                down = down.substr(0, down.size() - drop);
                piece_len.pop_back();
                piece_gt.pop_back();
                piece_start.pop_back();
                piece_end.pop_back();
                continue;
            }
            break;
        }
        if (seq.empty()) continue;
        piece_start.push_back(n_start);
        piece_end.push_back(n_end);
        const int64_t remaining = (int64_t)want - (int64_t)down.size();
        if ((int64_t)seq.size() >= remaining) {
            down.append(seq, 0, remaining);
            piece_len.push_back((uint32_t)remaining);
        } else {
            down.append(seq);
            piece_len.push_back((uint32_t)seq.size());
        }
        prev_gt = gt;
        piece_gt.push_back(gt);
    }
    return {up, down};
}

}  // namespace vgh
