// main_hip.cpp -- `varigraph_hip genotype --load-graph G -s S [options] [--gpu N] [--buffer MiB]`
//
// Entry point of the integration build (reference sources + this repo's adapter, see
// varigraph_hip.hpp).  Same `genotype` options, defaults and checks as the reference's CLI
// (main.cpp:238-408), plus the two device flags its CUDA build has (main.cu:99-100,302-303).
#include <getopt.h>

#include <cstdio>
#include <iostream>
#include <string>

#include "varigraph_hip.hpp"

static void usage(const char* argv0) {
    std::cerr << "Usage: " << argv0 << " genotype --load-graph FILE -s FILE [options]\n"
              << "  -g, --genotype hom|het [het]   --sample-ploidy INT [2]   -n, --number INT [15]\n"
              << "  --granularity FLOAT(Mb) [1]    -m, --mode fre|rec [rec]  --sv   --min-support FLOAT [0]\n"
              << "  --use-depth   -D, --debug   -t, --threads INT [10]\n"
              << "  --gpu INT [0]   --buffer INT(MiB) [100]\n";
}

int main(int argc, char** argv) {
    if (argc < 2 || std::string(argv[1]) != "genotype") { usage(argv[0]); return 1; }
    VarigraphConfig cfg;  // defaults: include/varigraph.hpp:49-68
    int gpu = 0;
    long buffer = 100;
    enum { O_LOAD = 1, O_PLOIDY, O_GRAN, O_SV, O_MINSUP, O_DEPTH, O_GPU, O_BUFFER };
    static const option opts[] = {
        {"load-graph", required_argument, 0, O_LOAD}, {"sample", required_argument, 0, 's'},
        {"genotype", required_argument, 0, 'g'},      {"sample-ploidy", required_argument, 0, O_PLOIDY},
        {"number", required_argument, 0, 'n'},        {"granularity", required_argument, 0, O_GRAN},
        {"mode", required_argument, 0, 'm'},          {"sv", no_argument, 0, O_SV},
        {"min-support", required_argument, 0, O_MINSUP}, {"use-depth", no_argument, 0, O_DEPTH},
        {"debug", no_argument, 0, 'D'},               {"threads", required_argument, 0, 't'},
        {"gpu", required_argument, 0, O_GPU},         {"buffer", required_argument, 0, O_BUFFER},
        {"help", no_argument, 0, 'h'},                {0, 0, 0, 0}};
    optind = 2;
    for (int c; (c = getopt_long(argc, argv, "s:g:n:m:Dt:h", opts, nullptr)) != -1;) {
        switch (c) {
            case O_LOAD: cfg.inputGraphFileName = optarg; break;
            case 's': cfg.samplesConfigFileName = optarg; break;
            case 'g': cfg.sampleType = optarg; break;
            case O_PLOIDY: cfg.samplePloidy = std::max(std::stoi(optarg), 2); break;
            case 'n': cfg.haploidNum = std::stoull(optarg); break;
            case O_GRAN: cfg.chrLenThread = std::stof(optarg) * 1e6; break;
            case 'm': cfg.transitionProType = optarg; break;
            case O_SV: cfg.svGenotypeBool = true; break;
            case O_MINSUP: cfg.minSupportingGQ = std::stof(optarg); break;
            case O_DEPTH: cfg.useDepth = true; break;
            case 'D': cfg.debug = true; break;
            case 't': cfg.threads = std::max(std::stoi(optarg), 1); break;
            case O_GPU: gpu = std::stoi(optarg); break;
            case O_BUFFER: buffer = std::stol(optarg); break;
            default: usage(argv[0]); return 1;
        }
    }
    auto bad = [&](const char* msg) { std::cerr << "Parameter error: " << msg << "\n\n"; usage(argv[0]); return 1; };
    if (cfg.inputGraphFileName.empty()) return bad("--load-graph");
    if (cfg.samplesConfigFileName.empty()) return bad("-s");
    if (cfg.sampleType != "hom" && cfg.sampleType != "het") return bad("-g must be hom or het");
    if (cfg.samplePloidy == 0 || cfg.samplePloidy > 8) return bad("--sample-ploidy must be 2..8");
    if (cfg.haploidNum == 0) return bad("-n must be > 0");
    if (cfg.chrLenThread < 1) return bad("--granularity");
    if (cfg.transitionProType != "fre" && cfg.transitionProType != "rec") return bad("-m must be fre or rec");
    if (gpu < 0) return bad("--gpu");
    if (buffer < 1) return bad("--buffer");

    cfg.logGenotypeConfig();
    VarigraphHip vg(cfg, gpu, (size_t)buffer);
    vg.parse_sample_config();
    vg.load();
    vg.fastq_genotype_hip();
    return 0;
}
