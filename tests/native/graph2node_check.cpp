// GraphIndex::graph2node through its batched lookup hook (what the CLI points at vgmi_table_lookup) against its own host index:
// the node lists must be the same arrays.  Usage: graph2node_check graph.bin [threads]
#include <cstdio>
#include <cstdlib>
#include <unordered_map>

#include "graph_index.hpp"

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const unsigned threads = argc > 2 ? (unsigned)atoi(argv[2]) : 4;
    vgh::GraphIndex a, b, c;
    a.threads = b.threads = c.threads = threads;
    size_t asked = 0;
    a.batched_find = [&](const uint64_t*, size_t n, uint32_t*) {      // declines, but sees how many k-mers the nodes hold
        asked = n;
        return false;
    };
    a.load(argv[1]);
    size_t calls = 0;
    b.batched_find = [&](const uint64_t* keys, size_t n, uint32_t* out) {
        ++calls;
        std::unordered_map<uint64_t, uint32_t> at;
        for (size_t i = 0; i < b.keys.size(); ++i) at.emplace(b.keys[i], (uint32_t)i);
        for (size_t i = 0; i < n; ++i) {
            auto it = at.find(keys[i]);
            out[i] = it == at.end() ? 0xFFFFFFFFu : it->second;
        }
        return true;
    };
    b.load(argv[1]);
    c.batched_find = [](const uint64_t*, size_t, uint32_t*) { return false; };      // cannot serve: the host index takes over
    c.load(argv[1]);
    size_t over = 0, absent = 0;
    for (size_t v = 0; v + 1 < a.node_off.size(); ++v) over += a.node_off[v + 1] - a.node_off[v] == 128;
    absent = asked - a.node_key_index.size();
    for (const vgh::GraphIndex* g : {&b, &c}) {
        if (g->node_off != a.node_off || g->node_key_index != a.node_key_index || g->node_start != a.node_start || g->node_chr != a.node_chr ||
            g->chr_names != a.chr_names || g->hom_flag != a.hom_flag) {
            std::printf("DIFFERENT\n");
            return 1;
        }
    }
    std::printf("identical: %zu nodes, %zu entries, %zu k-mers asked, %zu dropped, %zu nodes of 128 entries, %zu batched calls\n", a.node_off.size() - 1,
                a.node_key_index.size(), asked, absent, over, calls);
    return 0;
}
