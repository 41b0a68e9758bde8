// main_genotype.cpp -- `varigraph-mi genotype`: the reference's `varigraph genotype` sub-command (main.cpp:238-408,
// src/varigraph.cpp:104-243) on the MI355X build: graph.bin -> device table, FASTQ -> device k-mer counting, host HMM,
// <sample>.varigraph.vcf.gz in the working directory.  Same options (plus --gpu / --buffer as main.cu:99-100,302-303),
// same output naming, print-and-exit error behaviour.
#include <getopt.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <random>
#include <thread>
#include <set>
#include <unistd.h>
#include <signal.h>
#include <sys/wait.h>
#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "constructor.hpp"
#include "fastq_kmer_hip.hpp"
#include "genotyper.hpp"
#include "graph_index.hpp"
#include "vgmi.h"

namespace {

struct Options {
    std::string graph = "graph.bin", samples;
    vgh::GenotypeConfig hmm;
    bool use_depth = false;
    std::vector<int> gpus = {0};   // --gpu N, or --gpus a,b,c: samples are dealt to the devices, one
                                   // counting thread and one table replica per device
    int buffer_mib = 100;   // main.cu default
    bool procs = false;     // --procs: one PROCESS per device of --gpus, the table image handed on by one RCCL broadcast
};

// --procs: what the ranks share (one page, mapped before the fork): rank 0's RCCL id, whether it is there, whether anyone gave up
struct ProcShared {
    char nccl_id[128];
    std::atomic<int> id_ready;
    std::atomic<int> failed;
};
ProcShared* g_shared = nullptr;

void usage(const char* argv0)
{
    std::cerr << "Usage: " << argv0 << " genotype --load-graph FILE -s FILE [options]\n"
              << "Perform genotyping based on k-mer counting (MI355X build).\n\n"
              << "    --load-graph    FILE   Genome Graph index written by `varigraph construct` [graph.bin]\n"
              << "    -s, --sample    FILE   samples configuration: sample read1.fq.gz [read2.fq.gz ...] per line\n"
              << "    -g, --genotype  STR    sample genome status, hom/het [het]\n"
              << "    --sample-ploidy INT    sample ploidy, 2-8 [2]\n"
              << "    -n, --number    INT    haploids used for genotyping [15]\n"
              << "    --granularity   FLOAT  chromosome granularity in Mb [1]\n"
              << "    -m, --mode      STR    transition probabilities, fre/rec [rec]\n"
              << "    --sv                   structural variants only\n"
              << "    --min-support   FLOAT  minimum site quality (GQ) [0]\n"
              << "    --use-depth            use the sequencing depth as the depth of homozygous k-mers\n"
              << "    --gpu           INT    device ordinal [0]\n"
              << "    --gpus          LIST   several devices, e.g. 0,1,2,3: samples are counted on them in parallel\n"
              << "    --procs                one process per device of --gpus (samples dealt round robin, -t shared out); the table is\n"
              << "                           built on the first device and reaches the others by one RCCL broadcast\n"
              << "    --buffer        INT    staging buffer in MiB [100]\n"
              << "    -D, --debug            single host thread, phase times on stderr\n"
              << "    -t, --threads   INT    host threads [10]\n";
}

[[noreturn]] void die(const std::string& msg)
{
    if (g_shared) g_shared->failed = 1;       // ranks waiting for this one stop waiting
    std::cerr << "[varigraph-mi] " << msg << std::endl;
    std::fflush(nullptr);
    std::_Exit(1);   // other threads (parsers, device streams) may be mid-flight: no static destructors under them
}

// option values: the reference hands optarg to std::stoi / stof / stoull bare (main.cpp:127-144,294-320), so a
// non-numeric value ends it in std::terminate; here it is a parameter error like the range checks
int opt_int(const char* name, const char* v)
{
    try {
        size_t used = 0;
        const int r = std::stoi(v, &used);
        if (used == 0) throw std::invalid_argument(v);
        return r;
    } catch (const std::exception&) {
        die(std::string("Parameter error: ") + name + ". '" + v + "' is not an integer.");
    }
}
unsigned long long opt_u64(const char* name, const char* v)
{
    try {
        if (v[0] == '-') throw std::invalid_argument(v);
        return std::stoull(v);
    } catch (const std::exception&) {
        die(std::string("Parameter error: ") + name + ". '" + v + "' is not a non-negative integer.");
    }
}
float opt_float(const char* name, const char* v)
{
    try {
        return std::stof(v);
    } catch (const std::exception&) {
        die(std::string("Parameter error: ") + name + ". '" + v + "' is not a number.");
    }
}

// Varigraph::parse_sample_config (src/varigraph.cpp:104-146)
std::vector<std::tuple<std::string, std::vector<std::string>>> parse_samples(const std::string& path)
{
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) die("'" + path + "': No such file or directory.");
    std::string all;
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof buf)) > 0) all.append(buf, n);
    gzclose(f);
    std::vector<std::tuple<std::string, std::vector<std::string>>> out;
    std::istringstream in(all);
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::istringstream iss(line);
        std::vector<std::string> tok;
        for (std::string t; iss >> t;) tok.push_back(t);
        if (tok.empty()) continue;
        if (tok.size() <= 1) die("Error: The samples configuration file is missing sequencing file information (" + line + ").");
        std::vector<std::string> files;
        for (size_t i = 1; i < tok.size(); i++) {
            std::error_code ec;
            if (std::filesystem::exists(tok[i], ec) && std::filesystem::file_size(tok[i], ec) > 0) files.push_back(tok[i]);
            else die("Error: File '" + tok[i] + "' does not exist or is empty.");
        }
        out.emplace_back(tok[0], files);
    }
    return out;
}

int main_genotype(int argc, char** argv)
{
    Options o;
    bool debug = false;
    static const struct option long_options[] = {
        {"load-graph", required_argument, 0, 1},   {"sample", required_argument, 0, 's'},
        {"genotype", required_argument, 0, 'g'},   {"sample-ploidy", required_argument, 0, 2},
        {"number", required_argument, 0, 'n'},     {"granularity", required_argument, 0, 3},
        {"mode", required_argument, 0, 'm'},       {"sv", no_argument, 0, 4},
        {"min-support", required_argument, 0, 5},  {"use-depth", no_argument, 0, 6},
        {"gpu", required_argument, 0, 7},          {"buffer", required_argument, 0, 8},
        {"gpus", required_argument, 0, 9},         {"debug", no_argument, 0, 'D'},
        {"threads", required_argument, 0, 't'},    {"help", no_argument, 0, 'h'},
        {"procs", no_argument, 0, 10},
        {0, 0, 0, 0}};
    for (;;) {
        int idx = 0;
        const int c = getopt_long(argc, argv, "s:g:n:m:t:Dh", long_options, &idx);
        if (c == -1) break;
        switch (c) {
        case 1: o.graph = optarg; break;
        case 's': o.samples = optarg; break;
        case 'g': o.hmm.sample_type = optarg; break;
        case 2: o.hmm.sample_ploidy = std::max(opt_int("--sample-ploidy", optarg), 2); break;
        case 'n': o.hmm.haploid_num = (uint32_t)opt_u64("-n", optarg); break;
        case 3: o.hmm.chr_len_thread = opt_float("--granularity", optarg) * 1e6; break;
        case 'm': o.hmm.transition = optarg; break;
        case 4: o.hmm.sv_only = true; break;
        case 5: o.hmm.min_gq = opt_float("--min-support", optarg); break;
        case 6: o.use_depth = true; break;
        case 7: o.gpus = {opt_int("--gpu", optarg)}; break;
        case 9: {
            o.gpus.clear();
            std::stringstream ss(optarg);
            for (std::string t; std::getline(ss, t, ',');)
                if (!t.empty()) o.gpus.push_back(opt_int("--gpus", t.c_str()));
            if (o.gpus.empty()) die("Parameter error: --gpus. Expected a comma-separated list of device ordinals.");
            break;
        }
        case 8: o.buffer_mib = opt_int("--buffer", optarg); break;
        case 10: o.procs = true; break;
        case 't': o.hmm.threads = std::max(opt_int("-t", optarg), 1); break;
        case 'D': debug = true; break;
        default: usage(argv[0]); return 1;
        }
    }
    if (debug) {   // main.cpp:317, genotype.cpp:63-65: debug runs single-threaded; here it also prints the phase times
        o.hmm.threads = 1;
        setenv("VGH_TIMING", "1", 1);
    }
    if (o.graph.empty()) die("Parameter error: --load-graph. The genome graph file cannot be empty.");
    if (o.samples.empty()) die("Parameter error: -s. The sample configuration file cannot be empty.");
    if (o.hmm.sample_type != "hom" && o.hmm.sample_type != "het") die("Parameter error: -g. The provided value must be either 'hom' or 'het'.");
    if (o.hmm.sample_ploidy == 0 || o.hmm.sample_ploidy > 8) die("Parameter error: --sample-ploidy. The provided value must be between 2 and 8 (inclusive).");
    if (o.hmm.haploid_num == 0) die("Parameter error: -n. The provided value must be greater than 0.");
    if (o.hmm.haploid_num < 10)   // main.cpp:367-369
        std::cerr << "[varigraph-mi] Parameter warning: -n. The number of haploids for genotyping is relatively low, which may affect the accuracy of genotyping.\n";
    if (o.hmm.chr_len_thread < 1) die("Parameter error: --granularity. The chromosome granularity must be greater than 1.");
    if (o.hmm.chr_len_thread < 1000)   // main.cpp:375-377
        std::cerr << "[varigraph-mi] Parameter warning: --granularity. The chromosome granularity is less than 1000bp (" << o.hmm.chr_len_thread << " bp).\n";
    if (o.hmm.transition != "fre" && o.hmm.transition != "rec") die("Parameter error: -m. The transition probability type must be either 'fre' or 'rec'.");
    if (o.buffer_mib < 1) die("Parameter error: --buffer. The buffer size must be at least 1 MiB.");

    const auto t0 = std::chrono::steady_clock::now();
    auto secs = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    // The runtime gives a process four hardware queues by default and lets the streams beyond them share one: a sample's counting
    // then waits behind another sample's recursion kernel for its whole length (0.4-0.6 s at chr20 scale; with eight samples in
    // flight the later samples' counting took 0.6-0.75 s instead of 0.1, and the run two rounds instead of one:
    // profiles/r5_c4_stages.txt).  Asked for before anything touches the device; a value the user has set stays.
    setenv("GPU_MAX_HW_QUEUES", "24", 0);

    auto samples = parse_samples(o.samples);   // exits on a bad list before anything touches the device
    vgh::GraphIndex g;
    // --procs: graph.bin is parsed ONCE, here, by the parent and before anything touches a device (a process that has initialised
    // the GPU must not fork): the ranks share the parsed graph copy-on-write -- one copy of its 25 GB at whole-genome scale, not
    // one per rank -- and none of them reads the file again.  The node lists are resolved against the host's own key index (the
    // device's batched lookup serves the one-process run, below).  Then the ranks are forked.  Rank r takes device gpus[r], the
    // samples r, r + n, ... of the list (independent units, src/varigraph.cpp:153-172) and its share of -t; rank 0 builds the
    // table, every rank receives it in ONE ncclBroadcast -- unless the list names a device twice (RCCL takes one rank per
    // device): then every rank builds its own.  The communicator (seconds of ncclCommInitRank) comes up beside rank 0's table
    // build: rank 0 makes the RCCL id as the first thing it does, every rank starts ncclCommInitRank as soon as it sees it.
    int proc_rank = -1, proc_world = 0;
    bool proc_bcast = false;
    bool graph_loaded = false;
    if (o.procs) {
        const size_t n = o.gpus.size();
        g_shared = static_cast<ProcShared*>(mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
        if (g_shared == MAP_FAILED) { g_shared = nullptr; die("--procs: cannot map the shared page"); }
        new (g_shared) ProcShared{};
        // (one rank has nobody to send to: no communicator, unless VGH_PROCS_RCCL=1 asks for the communicator of one -- the test of the path on a single device)
        const bool rccl_of_one = [] { const char* e = getenv("VGH_PROCS_RCCL"); return e && e[0] == '1'; }();
        proc_bcast = std::set<int>(o.gpus.begin(), o.gpus.end()).size() == n && (n > 1 || rccl_of_one);
        try {
            g.threads = std::max(1u, o.hmm.threads);
            g.load(o.graph);
            if (samples.size() > n) g.build_entry_words();      // ranks with several samples run several Genotypers: their shared half, shared by all
        } catch (const std::exception& e) {
            die(e.what());
        }
        graph_loaded = true;
        std::cerr << "[varigraph-mi] graph parsed once for " << n << " ranks: " << g.keys.size() << " k-mers (" << secs() << " s)" << std::endl;
        std::fflush(nullptr);
        std::vector<pid_t> kids;
        for (size_t r = 0; r < n && proc_rank < 0; ++r) {
            const pid_t pid = fork();
            if (pid < 0) die("--procs: fork failed");
            if (pid == 0) proc_rank = (int)r;
            else kids.push_back(pid);
        }
        if (proc_rank < 0) {          // the parent only waits -- and ends the others when one rank gives up (they may be inside a collective)
            int worst = 0;
            size_t left = kids.size();
            while (left) {
                int st = 0;
                const pid_t k = waitpid(-1, &st, 0);
                if (k < 0) {
                    if (errno == EINTR) continue;
                    break;
                }
                if (std::find(kids.begin(), kids.end(), k) == kids.end()) continue;
                --left;
                const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
                if (rc != 0 && worst == 0) {
                    g_shared->failed = 1;
                    for (pid_t other : kids)
                        if (other != k) (void)kill(other, SIGTERM);
                }
                worst = std::max(worst, rc);
            }
            std::_Exit(worst);
        }
        proc_world = (int)n;
        std::vector<std::tuple<std::string, std::vector<std::string>>> mine;
        for (size_t i = (size_t)proc_rank; i < samples.size(); i += n) mine.push_back(samples[i]);
        samples.swap(mine);
        o.gpus = {o.gpus[(size_t)proc_rank]};
        o.hmm.threads = std::max<uint32_t>(1, o.hmm.threads / (uint32_t)n);
        g.threads = o.hmm.threads;
        std::cerr << "[varigraph-mi] rank " << proc_rank << " of " << proc_world << ": device " << o.gpus[0] << ", " << samples.size()
                  << " samples, " << o.hmm.threads << " threads"
                  << (proc_bcast ? "" : proc_world == 1 ? " (one rank: no broadcast)" : " (a device is named twice: every rank builds its table)") << std::endl;
    }
    std::cerr << "[varigraph-mi] samples: " << samples.size() << ", graph: " << o.graph << ", devices: " << o.gpus.size() << std::endl;
    // the device contexts come up (HIP runtime start, staging buffers) while the graph is read
    std::vector<vgmi_ctx*> ctxs;
    std::string ctx_error;
    // ... and the first device builds its table as soon as the k-mer records are read, while the host still resolves the
    // node lists (graph2node)
    std::mutex keys_mu;
    std::condition_variable keys_cv;
    int keys_state = graph_loaded ? 1 : 0;   // 1: keys complete, -1: the load failed before that
    int table_state = 0;  // 1: the first device holds the table, -1: it never will
    if (!graph_loaded) {
        g.on_keys = [&] {
            {
                std::lock_guard<std::mutex> lk(keys_mu);
                keys_state = 1;
            }
            keys_cv.notify_all();
        };
        // graph2node's lookups (every k-mer of every variant node against the key set) go to that table as one batch
        g.batched_find = [&](const uint64_t* keys, size_t n, uint32_t* index_out) {
            if (const char* e = getenv("VGH_DEVICE_GRAPH2NODE"))
                if (e[0] == '0') return false;
            {
                std::unique_lock<std::mutex> lk(keys_mu);
                keys_cv.wait(lk, [&] { return table_state != 0; });
                if (table_state < 0) return false;
            }
            return vgmi_table_lookup(ctxs[0], keys, n, index_out) == VGMI_OK;
        };
    }
    // The buffers of a sample's two FASTQ streams (1.5 GB of pinned staging and device text at the default --buffer) are allocated
    // while the graph still loads, not inside the first sample's counting (0.3 s for 12 M pairs, 0.15 s of it allocation): a stream
    // that is closed leaves its buffers in the context's pool
    auto warm_fastq = [&](vgmi_ctx* ctx) {
        if (!(g.k & 1) || samples.empty()) return;
        vgmi_fastq* fq[2] = {nullptr, nullptr};
        const size_t n_streams = std::min<size_t>(2, std::get<1>(samples[0]).size());
        for (size_t i = 0; i < n_streams; ++i)
            if (vgmi_fastq_open(ctx, &fq[i]) != VGMI_OK) fq[i] = nullptr;
        char tail[8];
        size_t tail_len = 0;
        for (size_t i = 0; i < n_streams; ++i)
            if (fq[i]) (void)vgmi_fastq_close(fq[i], nullptr, nullptr, nullptr, nullptr, tail, sizeof tail, &tail_len);
    };
    // --procs: the communicator, on a thread of its own from the moment the id is there
    vgmi_comm* comm = nullptr;
    std::string comm_error;
    double comm_seconds = 0;
    std::thread comm_up, bcast_thr;
    if (proc_bcast) {
        if (proc_rank == 0) {
            if (vgmi_rccl_unique_id(g_shared->nccl_id) != VGMI_OK) die(vgmi_last_error(nullptr));
            g_shared->id_ready = 1;
        }
        comm_up = std::thread([&] {
            while (!g_shared->id_ready.load()) {
                if (g_shared->failed.load() || getppid() == 1) { comm_error = "--procs: another rank gave up"; return; }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            const auto t_c = std::chrono::steady_clock::now();
            if (vgmi_comm_create(o.gpus[0], proc_rank, proc_world, g_shared->nccl_id, &comm) != VGMI_OK) comm_error = vgmi_last_error(nullptr);
            comm_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_c).count();
        });
    }
    std::thread bring_up([&] {
        struct TableFlag {      // whatever way this thread ends, a waiting graph2node hears of it
            std::mutex& mu; std::condition_variable& cv; int& state;
            ~TableFlag() { { std::lock_guard<std::mutex> lk(mu); if (state == 0) state = -1; } cv.notify_all(); }
        } table_flag{keys_mu, keys_cv, table_state};
        for (int dev : o.gpus) {
            vgmi_ctx* ctx = nullptr;
            if (vgmi_create(dev, (size_t)o.buffer_mib, &ctx) != VGMI_OK) {
                ctx_error = std::string("device ") + std::to_string(dev) + ": " + vgmi_last_error(nullptr);
                return;
            }
            ctxs.push_back(ctx);
        }
        auto table_there = [&] {
            {
                std::lock_guard<std::mutex> lk2(keys_mu);
                table_state = 1;
            }
            keys_cv.notify_all();
            warm_fastq(ctxs[0]);
        };
        if (proc_bcast && proc_rank > 0) {
            // the table arrives from rank 0 (no key upload, no build here): the communicator, then the broadcast
            const auto t_wait = std::chrono::steady_clock::now();
            comm_up.join();
            if (!comm_error.empty()) { ctx_error = comm_error; return; }
            if (vgmi_table_broadcast_comm(ctxs[0], comm) != VGMI_OK) {
                ctx_error = vgmi_last_error(ctxs[0]);
                return;
            }
            std::fprintf(stderr, "[varigraph-mi] rank %d: table image received by RCCL broadcast %.3f s after the context was up (communicator %.3f s)\n",
                         proc_rank, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count(), comm_seconds);
            table_there();
            return;
        }
        std::unique_lock<std::mutex> lk(keys_mu);
        keys_cv.wait(lk, [&] { return keys_state != 0; });
        if (keys_state < 0) return;
        lk.unlock();
        // ONE table build (first device)
        if (vgmi_table_upload(ctxs[0], g.keys.data(), g.keys.size(), g.k) != VGMI_OK) {
            ctx_error = vgmi_last_error(ctxs[0]);
            // (the other ranks hear of it inside the broadcast: size 0)
            if (proc_bcast && proc_rank == 0) {
                comm_up.join();
                if (comm) (void)vgmi_table_broadcast_comm(ctxs[0], comm);
            }
            return;
        }
        if (proc_bcast && proc_rank == 0) {
            // the others' copy leaves from a snapshot of the image, whenever the communicator is up: this rank does not wait for it
            if (vgmi_table_snapshot(ctxs[0]) != VGMI_OK) {
                // no room for a second copy of the image: the broadcast leaves from the image itself, before this rank counts (the
                // other ranks are inside the collective either way -- returning here left them there)
                comm_up.join();
                if (!comm_error.empty()) { ctx_error = comm_error; return; }
                if (vgmi_table_broadcast_comm(ctxs[0], comm) != VGMI_OK) { ctx_error = vgmi_last_error(ctxs[0]); return; }
                std::fprintf(stderr, "[varigraph-mi] rank 0: no memory for a snapshot of the table image: sent to %d ranks before counting\n", proc_world - 1);
                table_there();
                return;
            }
            bcast_thr = std::thread([&] {
                const auto t_wait = std::chrono::steady_clock::now();
                comm_up.join();
                if (!comm_error.empty()) die(comm_error);
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count();
                if (vgmi_table_broadcast_comm(ctxs[0], comm) != VGMI_OK) die(vgmi_last_error(ctxs[0]));
                size_t ib = 0;
                (void)vgmi_table_image_bytes(ctxs[0], &ib);
                std::fprintf(stderr, "[varigraph-mi] rank 0: table image of %.1f MB sent to %d ranks in one RCCL broadcast (communicator %.3f s beside the table build, "
                                     "%.3f s of it after the table was there, broadcast %.3f s; rank 0 counted meanwhile)\n", ib / 1e6, proc_world - 1,
                             comm_seconds, waited, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count() - waited);
            });
        }
        table_there();
    });
    std::string load_error;
    if (!graph_loaded) {
        try {
            g.threads = std::max(1u, o.hmm.threads);
            g.load(o.graph);
        } catch (const std::exception& e) {
            load_error = e.what();
        }
    }
    {
        std::lock_guard<std::mutex> lk(keys_mu);
        if (keys_state == 0) keys_state = -1;
    }
    keys_cv.notify_all();
    bring_up.join();
    if (!bcast_thr.joinable() && comm_up.joinable()) comm_up.join();      // (rank 0's broadcast thread joins it itself)
    if (!load_error.empty()) die(load_error);
    if (!ctx_error.empty()) die(ctx_error);
    // the image then goes device to device in a doubling tree (round r: the 2^r devices that hold it feed the next 2^r over
    // xGMI), never through the host again; node lists and flags are small host uploads
    if (g.upload_nodes(ctxs[0]) != VGMI_OK) die(vgmi_last_error(ctxs[0]));
    const double t_built = secs();
    size_t image_bytes = 0;
    (void)vgmi_table_image_bytes(ctxs[0], &image_bytes);
    for (size_t have = 1; have < ctxs.size(); have *= 2) {
        std::vector<std::thread> copies;
        std::vector<std::string> errs(ctxs.size());
        for (size_t i = 0; i < have && have + i < ctxs.size(); ++i)
            copies.emplace_back([&, i] {
                vgmi_ctx* dst = ctxs[have + i];
                if (vgmi_table_clone(dst, ctxs[i]) != VGMI_OK || g.upload_nodes(dst) != VGMI_OK) errs[have + i] = vgmi_last_error(dst);
                else warm_fastq(dst);
            });
        for (auto& t : copies) t.join();
        for (const auto& e : errs)
            if (!e.empty()) die(e);
    }
    if (ctxs.size() > 1)
        std::fprintf(stderr, "[varigraph-mi] table: 1 build on device %d, %zu device-to-device image copies of %.1f MB (%.3f s)\n", o.gpus[0],
                     ctxs.size() - 1, image_bytes / 1e6, secs() - t_built);
    std::cerr << "[varigraph-mi] graph loaded: " << g.keys.size() << " k-mers, k = " << g.k << ", " << g.hap_names.size()
              << " haplotypes (" << secs() << " s)" << std::endl;

    // Two stages.  Counting: one thread per device takes the next sample of the `-s` list (samples are independent
    // units).  Genotyping: one consumer, strictly in `-s` order, when the HMM keeps state across samples (the
    // pruned node lists), exactly like Varigraph::fastq_genotype (src/varigraph.cpp:148-171), which runs the two
    // stages back to back -- several consumers when it does not (below); either way the counting of later samples
    // overlaps the HMM of earlier ones.
    struct Job {
        std::string name;
        std::vector<uint8_t> cov_node;     // the sample's counters gathered per node on the device (K5): what the HMM's windows walk
        float hap_cov = 0;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::map<size_t, Job> ready;           // finished counting, keyed by position in the -s list
    size_t next_to_genotype = 0;
    std::atomic<size_t> next_to_count{0};
    const size_t max_ready = 2 * ctxs.size();   // samples counted but not yet taken by a genotyping consumer

    // The HMM prunes a node's k-mer list to the k-mers some selected haplotype carries and the pruned list stays for
    // the next sample (genotype.cpp:815-818), so samples must be genotyped in order by one consumer -- unless nothing
    // can be pruned: every haplotype is selected (-n >= #haplotypes) and every k-mer is carried by one.  Then the
    // samples are independent and the consumers (one per device, two on a single one) run them side by side, each on its own
    // Genotyper.
    // With the HMM's recursion on the device a consumer's threads idle while its chains run (a chain is latency, §4.8 of
    // DESIGN.md): two consumers share one device's worth of samples, each with half the threads.
    const bool device_hmm = [] { const char* e = getenv("VGH_HMM_DEVICE"); return !(e && e[0] == '0'); }();
    // (a consumer holds the nodes' k-mer lists and a packed word per k-mer of its own: 12 bytes per k-mer -- not doubled for a
    // graph of more than 2^29 k-mers)
    const bool second_consumer = device_hmm && g.keys.size() < ((size_t)1 << 29);
    // ... four on a single device while the graph is small enough for four copies of that: with the threads on one budget
    // (below) a consumer's host phases take as long as they would alone, and the device runs the chains of four samples side by
    // side at little more than one sample's latency (tools/gpu_hmm_pack.sh: 960 chains in 39.5 ms, 60 chains in 33.5 ms);
    // eight chr20-scale samples, -t 10: 6.1 s with two consumers on fixed shares, 5.7 s on the budget, 4.5-5.1 s with four
    const bool four = second_consumer && ctxs.size() == 1 && g.keys.size() < ((size_t)1 << 27);
    // ... eight where there are eight samples or more (round 5): a sample's device time is the latency of its chains -- 0.45-0.55 s at
    // chr20 scale whether one sample's 120 chains or eight samples' 960 are on the device -- and four consumers took eight samples in
    // two rounds of it (4.3-4.5 s, profiles/r5_c4_stages.txt)
    const size_t eight = four && samples.size() >= 8 ? 8 : 4;
    size_t per_run = std::max<size_t>(ctxs.size(), four ? eight : second_consumer ? 2 : 1);
    if (const char* e = getenv("VGH_HMM_CONSUMERS")) per_run = (size_t)std::max(1L, atol(e));      // A/B
    const size_t want_consumers = std::max<size_t>(1, std::min(samples.size(), per_run));
    bool independent = g.hap_names.size() <= o.hmm.haploid_num;
    if (independent && want_consumers > 1) {
        const size_t bl = g.bitlen, n_hap = g.hap_names.size();
        for (size_t r = 0; r < g.keys.size() && independent; ++r) {
            bool any = false;
            for (size_t h = 0; h < n_hap && !any; ++h) any = ((uint8_t)g.bitvec[r * bl + (h >> 3)] >> (h & 7)) & 1u;
            independent = any;
        }
    }
    if (proc_world > 1 && !independent && g.hap_names.size() > o.hmm.haploid_num)
        die("--procs deals the samples to several processes: that needs samples that are independent units (-n >= the number of "
            "haplotypes of the graph, so that no sample's haplotype selection prunes the next one's k-mer lists)");
    const size_t n_consumers = independent ? want_consumers : 1;
    if (n_consumers > 1 && g.entry_words.empty()) g.build_entry_words();      // the graph's half of the Genotypers' per-entry words, once instead of once each
    // -t is the budget of the whole run: counting threads (inflate workers) and HMM consumers that run side by side
    // share it instead of each taking all of it
    const unsigned count_threads = std::max<unsigned>(1, o.hmm.threads / (unsigned)std::max<size_t>(1, std::min(ctxs.size(), samples.size())));
    vgh::GenotypeConfig hmm_cfg = o.hmm;
    // ... the consumers through one budget of running threads (vgh::CpuBudget): a consumer whose neighbours wait for the device has
    // all of -t for its own preparation, calls and text; VGH_CPU_BUDGET=0: a fixed share each (rounds 2-3)
    const bool shared_budget = n_consumers > 1 && [] { const char* e = getenv("VGH_CPU_BUDGET"); return !(e && e[0] == '0'); }();
    if (shared_budget) vgh::CpuBudget::set(std::max<unsigned>(1, o.hmm.threads));
    else hmm_cfg.threads = std::max<unsigned>(1, o.hmm.threads / (unsigned)n_consumers);
    std::atomic<size_t> next_hmm{0};
    std::atomic<unsigned> consumer_no{0};
    auto consumer = [&] {
        try {
            const double tc0 = secs();
            vgh::Genotyper genotyper(g, hmm_cfg.threads);
            const double tc1 = secs();
            // the HMM's recursion and posterior run on the device, through a context of their own (its stream and buffers are
            // not the counting's); VGH_HMM_DEVICE=0 keeps them on the host
            struct OwnCtx {
                vgmi_ctx* c = nullptr;
                ~OwnCtx() { if (c) vgmi_destroy(c); }
            } own;
            {
                const char* e = getenv("VGH_HMM_DEVICE");
                if (!(e && e[0] == '0')) {
                    const int dev = o.gpus[consumer_no.fetch_add(1) % o.gpus.size()];
                    if (vgmi_create(dev, 16, &own.c) != VGMI_OK) die(std::string("device ") + std::to_string(dev) + ": " + vgmi_last_error(nullptr));
                    size_t on_dev = 0;      // consumers whose HMM calls go to this device
                    for (size_t k = 0; k < n_consumers; ++k) on_dev += o.gpus[k % o.gpus.size()] == dev;
                    genotyper.set_device(own.c, (unsigned)std::max<size_t>(1, 4 / std::max<size_t>(1, on_dev)));
                }
            }
            if (getenv("VGH_TIMING"))
                std::fprintf(stderr, "[varigraph-mi] consumer: genotyper set up from %.2f to %.2f s, device context by %.2f s\n", tc0, tc1, secs());
            for (;;) {
                const size_t s = next_hmm.fetch_add(1);
                if (s >= samples.size()) return;
                Job job;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return ready.count(s) != 0; });
                    job = std::move(ready[s]);
                    ready.erase(s);
                    next_to_genotype = std::max(next_to_genotype, s + 1);
                }
                cv.notify_all();
                const double th = secs();
                const std::string vcf = genotyper.run(nullptr, job.hap_cov, job.name, hmm_cfg, job.cov_node.data());
                const double tz = secs();
                vgh::Genotyper::write_gz(job.name + ".varigraph.vcf.gz", vcf, hmm_cfg.threads);
                std::fprintf(stderr, "[varigraph-mi] %s: genotyping %.2f s (HMM %.2f with %zu of %zu windows on the device, VCF text %.2f, gzip %.2f) -> %s.varigraph.vcf.gz\n",
                             job.name.c_str(), secs() - th, genotyper.last_hmm_seconds, genotyper.last_device_windows, genotyper.last_windows,
                             genotyper.last_text_seconds, secs() - tz, job.name.c_str());
            }
        } catch (const std::exception& e) {
            die(e.what());
        }
    };
    std::vector<std::thread> hmm_threads;
    for (size_t c = 0; c < n_consumers; ++c) hmm_threads.emplace_back(consumer);
    // the read-out of a counted sample: the per-node depth lookup (src/genotype.cpp:546,660,1405-1408) and the masked histogram
    // (src/varigraph.cpp:253-296) happen on the device: node-ordered counters and 256 bins come back, the per-key array is not fetched
    auto counted = [&](size_t s, vgh::FastqKmerHip& fk, size_t dev_i, double ts) {
        const std::string& name = std::get<0>(samples[s]);
        Job job;
        job.name = name;
        job.cov_node.resize(g.node_key_index.size());
        uint64_t hist[256];
        const double tf = secs();
        fk.fetch(nullptr, job.cov_node.data(), hist);
        const double fetch_s = secs() - tf;
        vgh::CoverageStats cs;
        if (!vgh::coverage_stats(hist, fk.mReadBase, g.genome_size, o.hmm.sample_ploidy, o.use_depth, cs))
            die("Failed to retrieve depth information of k-mers from the sequencing data. Please verify your data.");
        job.hap_cov = cs.hap_kmer_coverage;
        std::fprintf(stderr, "[varigraph-mi] %s (device %d): %.2f Gb sequenced, depth %.2f, haplotype k-mer coverage %.2f; counting %.2f s (kernel %.3f s, read-out %.3f s)\n",
                     name.c_str(), o.gpus[dev_i], fk.mReadBase / 1e9, cs.read_depth, cs.hap_kmer_coverage, secs() - ts, fk.kernel_seconds(), fetch_s);
        {
            std::lock_guard<std::mutex> lk(mu);
            ready.emplace(s, std::move(job));
        }
        cv.notify_all();
    };
    auto counter = [&](size_t dev_i) {
        vgmi_ctx* ctx = ctxs[dev_i];
        try {
            for (;;) {
                const size_t s = next_to_count.fetch_add(1);
                if (s >= samples.size()) return;
                {   // do not run ahead of the HMM by more than a couple of samples per device (each holds n_keys bytes)
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return s < next_to_genotype + max_ready; });
                }
                const auto& [name, files] = samples[s];
                const double ts = secs();
                vgh::FastqKmerHip fk(ctx, files, g.k, count_threads);
                fk.build_fastq_index();
                counted(s, fk, dev_i, ts);
            }
        } catch (const std::exception& e) {
            die(e.what());
        }
    };
    std::vector<std::thread> counters;
    for (size_t d = 1; d < ctxs.size(); ++d) counters.emplace_back(counter, d);
    counter(0);
    for (auto& t : counters) t.join();
    for (auto& t : hmm_threads) t.join();
    if (bcast_thr.joinable()) bcast_thr.join();
    if (comm) { vgmi_comm_destroy(comm); comm = nullptr; }
    const double t_joined = secs();
    for (vgmi_ctx* ctx : ctxs) vgmi_destroy(ctx);
    if (getenv("VGH_TIMING")) std::fprintf(stderr, "[varigraph-mi] last sample written at %.2f s, devices released by %.2f s\n", t_joined, secs());
    {   // what this process held: its own peak (VmHWM counts pages shared copy-on-write in full) and its proportional share (Pss)
        double hwm = 0, pss = 0;
        if (FILE* f = std::fopen("/proc/self/status", "r")) {
            char line[256];
            while (std::fgets(line, sizeof line, f))
                if (std::strncmp(line, "VmHWM:", 6) == 0) hwm = std::atof(line + 6) / 1048576.0;
            std::fclose(f);
        }
        if (FILE* f = std::fopen("/proc/self/smaps_rollup", "r")) {
            char line[256];
            while (std::fgets(line, sizeof line, f))
                if (std::strncmp(line, "Pss:", 4) == 0) pss = std::atof(line + 4) / 1048576.0;
            std::fclose(f);
        }
        if (proc_rank >= 0) std::fprintf(stderr, "[varigraph-mi] rank %d: host memory: peak RSS %.3f GB, PSS at exit %.3f GB\n", proc_rank, hwm, pss);
        else std::fprintf(stderr, "[varigraph-mi] host memory: peak RSS %.3f GB, PSS at exit %.3f GB\n", hwm, pss);
    }
    std::fprintf(stderr, "[varigraph-mi] done in %.2f s\n", secs());
    // every output file is closed and the devices are released: skip taking the graph (1e7s of small allocations) apart
    std::fflush(nullptr);
    if (getenv("VGH_ATEXIT")) std::exit(0);      // (under a profiler: its library writes its trace from an exit handler, which _Exit skips)
    std::_Exit(0);
}

// `construct` (main.cpp:21-236, Varigraph::construct src/varigraph.cpp:14-54): same options plus --gpu / --buffer
int main_construct(int argc, char** argv)
{
    vgh::ConstructConfig c;
    int gpu = 0, buffer_mib = 100, vcf_ploidy = (int)c.vcf_ploidy, kmer = (int)c.k;
    bool debug = false;
    static const struct option long_options[] = {
        {"reference", required_argument, 0, 'r'}, {"vcf", required_argument, 0, 'v'},
        {"save-graph", required_argument, 0, 1},  {"vcf-ploidy", required_argument, 0, 2},
        {"kmer", required_argument, 0, 'k'},      {"fast", no_argument, 0, 3},
        {"use-unique-kmers", no_argument, 0, 4},  {"gpu", required_argument, 0, 7},
        {"buffer", required_argument, 0, 8},      {"threads", required_argument, 0, 't'},
        {"debug", no_argument, 0, 'D'},           {"help", no_argument, 0, 'h'},            {0, 0, 0, 0}};
    for (;;) {
        int idx = 0;
        const int o = getopt_long(argc, argv, "r:v:k:t:Dh", long_options, &idx);
        if (o == -1) break;
        switch (o) {
        case 'r': c.reference = optarg; break;
        case 'v': c.vcf = optarg; break;
        case 1: c.out = optarg; break;
        case 2: vcf_ploidy = std::max(opt_int("--vcf-ploidy", optarg), 2); break;   // main.cpp:127: max(stoi, 2)
        case 'k': kmer = std::max(opt_int("-k", optarg), 5); break;                  // main.cpp:131: max(stoi, 5)
        case 3: c.fast = true; break;
        case 4: c.use_unique_kmers = true; break;
        case 7: gpu = opt_int("--gpu", optarg); break;
        case 8: buffer_mib = opt_int("--buffer", optarg); break;
        case 't': c.threads = (uint32_t)std::max(opt_int("-t", optarg), 1); break;
        case 'D': debug = true; break;   // main.cpp:141, construct_index.cpp:33-35
        default:
            std::cerr << "Usage: construct -r FASTA -v VCF [--save-graph FILE] [-k INT] [--vcf-ploidy INT] [--fast] "
                         "[--use-unique-kmers] [--gpu INT] [--buffer INT]\n";
            return 1;
        }
    }
    if (c.reference.empty()) die("Parameter error: -r. The reference genome file cannot be empty.");
    if (c.vcf.empty()) die("Parameter error: -v. The VCF file cannot be empty.");
    if (c.out.empty()) die("Parameter error: --save-graph. The Genome Graph file cannot be empty.");
    // main.cpp:181-191
    if (vcf_ploidy <= 0 || vcf_ploidy > 8) die("Parameter error: --vcf-ploidy. The provided value must be between 2 and 8 (inclusive).");
    if (kmer <= 0 || kmer > 28) die("Parameter error: -k. The provided value must be between 1 and 28 (inclusive).");
    c.vcf_ploidy = (uint32_t)vcf_ploidy;
    c.k = (uint32_t)kmer;
    if (const char* e = std::getenv("VGH_RANDOM_DEVICE_VALUE")) {
        c.random_device_value = (uint32_t)std::strtoul(e, nullptr, 10);
    } else {
        std::random_device rd;
        c.random_device_value = rd();
    }
    const auto t0 = std::chrono::steady_clock::now();
    vgmi_ctx* ctx = nullptr;
    if (vgmi_create(gpu, (size_t)buffer_mib, &ctx) != VGMI_OK) die(std::string("device ") + std::to_string(gpu) + ": " + vgmi_last_error(nullptr));
    try {
        if (debug) {
            c.threads = 1;
            setenv("VGH_TIMING", "1", 1);
        }
        c.release_memory = false;   // the process ends right after
        const vgh::ConstructStats st = vgh::construct_graph(ctx, c);
        std::fprintf(stderr,
                     "[varigraph-mi] construct: genome %.2f Mb, %llu variant nodes, %llu k-mers, %llu haplotypes -> %s\n"
                     "[varigraph-mi]   Bloom %.1f MB on device in %.2f s, indexing %.2f s (%llu Bloom queries), total %.2f s\n",
                     st.genome_size / 1e6, (unsigned long long)st.n_variant_nodes, (unsigned long long)st.n_kmers,
                     (unsigned long long)st.n_haplotypes, c.out.c_str(), st.bloom_bytes / 1e6, st.seconds_bloom, st.seconds_index,
                     (unsigned long long)st.bloom_queries,
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    } catch (const std::exception& e) {
        die(e.what());
    }
    vgmi_destroy(ctx);
    std::fflush(nullptr);
    std::_Exit(0);   // graph.bin is closed; the containers were left alone on purpose (release_memory = false)
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 2 || std::string(argv[1]) == "-h" || std::string(argv[1]) == "--help") {
        usage(argv[0]);
        return 1;
    }
    const std::string cmd = argv[1];
    if (cmd == "genotype") {
        if (argc < 3) {
            usage(argv[0]);
            return 1;
        }
        return main_genotype(argc - 1, argv + 1);
    }
    if (cmd == "construct") {
        if (argc < 3) {
            std::cerr << "Usage: " << argv[0] << " construct -r FASTA -v VCF [--save-graph FILE] [options]\n";
            return 1;
        }
        return main_construct(argc - 1, argv + 1);
    }
    std::cerr << "Error: '" << cmd << "' is not a sub-command of this build (construct, genotype).\n";
    return 1;
}
