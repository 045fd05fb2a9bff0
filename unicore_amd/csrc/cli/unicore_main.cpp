// unicore_main.cpp — C++ host mirror of the reference's `unicore cluster` (and `unicore search`) module surface.
// (The reference host is Rust; no Rust toolchain exists in this image — SURVEY.md 0.2/D3 — so the host
// above the C ABI is C++ with the same names, argument meaning and error behaviour.)
//   CLI surface      /root/reference/src/util/arg_parser.rs:225-246  (Commands::Cluster)
//   module body      /root/reference/src/modules/cluster.rs:9-84     (modules::cluster::run)
//   checkpoint       /root/reference/src/util/checkpoint.rs:2-5
//   error/exit codes /root/reference/src/envs/error_handler.rs:5-45
//   messages         /root/reference/src/util/message.rs:4-22
// The three `foldseek` spawns of cluster.rs:45-76 become three in-process calls through the C ABI.
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "unicore_cluster.h"

namespace {

constexpr int ERR_GENERAL = 0x01, ERR_ARGPARSE = 0x40;   // error_handler.rs:6,13
int g_verbosity = 3;                                       // variables.rs:146

void print_message(const std::string &m, int level) { if (level <= g_verbosity) { fputs(m.c_str(), stdout); fflush(stdout); } }
void println_message(const std::string &m, int level) { if (level <= g_verbosity) { puts(m.c_str()); } }
[[noreturn]] void error(int code, const std::string &object) {   // error_handler.rs:42-45
    if (1 <= g_verbosity) fprintf(stderr, "%s%s\n", code == ERR_ARGPARSE ? "Argument parsing error: " : "Error: ", object.c_str());
    exit(code);
}

void write_checkpoint(const std::string &file, const char *content) {   // checkpoint.rs:2-5 (no newline)
    std::ofstream f(file, std::ios::binary | std::ios::trunc);
    if (!f) error(ERR_GENERAL, "Could not write checkpoint " + file);
    f << content;
}

void create_dir_all(const std::string &path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); i++) {
        if ((i == path.size() || path[i] == '/') && !cur.empty() && cur != "/") {
            struct stat st;
            if (stat(cur.c_str(), &st) != 0 && mkdir(cur.c_str(), 0777) != 0) error(ERR_GENERAL, "Could not create directory " + cur);
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
}

void usage(FILE *to) {
    fputs("Usage: unicore cluster [OPTIONS] <INPUT> <OUTPUT> <TMP>\n\n"
         "Arguments:\n"
         "  <INPUT>   Input database (createdb output)\n"
         "  <OUTPUT>  Output prefix; the result will be saved as OUTPUT.tsv\n"
         "  <TMP>     Temp directory\n\n"
         "Options:\n"
         "  -k, --keep-cluster-db            Keep intermediate cluster database\n"
         "  -c, --cluster-options <STRING>   Arguments for foldseek-style options in string e.g. -c \"-c 0.8\" [default: \"-c 0.8\"]\n"
         "      --threads <THREADS>          Number of threads to use; 0 to use all [default: 0]\n"
         "  -v, --verbosity <VERBOSITY>      Verbosity (0: quiet, 1: +errors, 2: +warnings, 3: +info, 4: +debug) [default: 3]\n"
         "  -h, --help                       Print help\n", to);
}

// modules::cluster::run (cluster.rs:9-84)
int cluster_run(const std::string &input, const std::string &output, const std::string &tmp, bool keep_cluster_db,
                const std::string &cluster_options, int threads) {
    const int engine_verbosity = g_verbosity == 4 ? 3 : g_verbosity == 3 ? 2 : g_verbosity;   // cluster.rs:18
    // parent directory of the output (cluster.rs:21-29).  Deviation: an OUTPUT without a directory
    // component yields "" in the reference (checkpoint lands in "/cluster.chk"); "." is the intent.
    size_t slash = output.find_last_of('/');
    std::string parent = slash == std::string::npos ? "." : (slash == 0 ? "/" : output.substr(0, slash));
    create_dir_all(parent);
    write_checkpoint(parent + "/cluster.chk", "0");   // cluster.rs:32

    const std::string output_cluster_db = output + "_cluster", output_tsv = output + ".tsv";   // cluster.rs:43-44
    uc_opts o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    o.threads = threads;
    o.verbosity = engine_verbosity;
    o.device = -1;
    o.num_gpus = 0;   // all visible GPUs; "--gpus N" inside the option string narrows it
    o.cluster_options = cluster_options.c_str();

    print_message("Running cluster on the MI355X engine...", 3);   // cluster.rs:52
    if (g_verbosity >= 3) putchar('\n');
    int rc = uc_cluster(input.c_str(), output_cluster_db.c_str(), tmp.c_str(), &o, nullptr);   // cluster.rs:45-55
    if (rc != 0) error(ERR_GENERAL, std::string("cluster engine failed with code ") + std::to_string(rc) + "\n" + uc_last_error());   // command.rs:10-14
    println_message(" Done", 3);                                    // cluster.rs:56
    rc = uc_createtsv(input.c_str(), output_cluster_db.c_str(), output_tsv.c_str(), &o);           // cluster.rs:59-64
    if (rc != 0) error(ERR_GENERAL, std::string("createtsv failed with code ") + std::to_string(rc) + "\n" + uc_last_error());
    if (!keep_cluster_db) {                                         // cluster.rs:67-76
        rc = uc_rmdb(output_cluster_db.c_str());
        if (rc != 0) error(ERR_GENERAL, std::string("rmdb failed with code ") + std::to_string(rc) + "\n" + uc_last_error());
    }
    write_checkpoint(parent + "/cluster.chk", "1");   // cluster.rs:81
    return 0;
}

void usage_search(FILE *to) {
    fputs("Usage: unicore search [OPTIONS] <INPUT> <TARGET> <OUTPUT> <TMP>\n\n"
         "Arguments:\n"
         "  <INPUT>   Input database\n"
         "  <TARGET>  Target database to search against\n"
         "  <OUTPUT>  Output prefix; the result will be saved as OUTPUT.m8\n"
         "  <TMP>     Temp directory\n\n"
         "Options:\n"
         "  -k, --keep-aln-db                Keep intermediate Foldseek alignment database\n"
         "  -s, --search-options <STRING>    Arguments for foldseek-style options in string e.g. -s \"-c 0.8\" [default: \"-c 0.8\"]\n"
         "      --threads <THREADS>          Number of threads to use; 0 to use all [default: 0]\n"
         "  -v, --verbosity <VERBOSITY>      Verbosity (0: quiet, 1: +errors, 2: +warnings, 3: +info, 4: +debug) [default: 3]\n"
         "  -h, --help                       Print help\n", to);
}

// modules::search::run (search.rs:8-84)
int search_run(const std::string &input, const std::string &target, const std::string &output, const std::string &tmp, bool keep_aln_db,
               const std::string &search_options, int threads) {
    const int engine_verbosity = g_verbosity == 4 ? 3 : g_verbosity == 3 ? 2 : g_verbosity;
    size_t slash = output.find_last_of('/');
    std::string parent = slash == std::string::npos ? "." : (slash == 0 ? "/" : output.substr(0, slash));   // search.rs:21-29
    create_dir_all(parent);
    write_checkpoint(parent + "/search.chk", "0");   // search.rs:32
    const std::string output_aln_db = output + "_aln", output_m8 = output + ".m8";   // search.rs:43-44
    uc_opts o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    o.threads = threads;
    o.verbosity = engine_verbosity;
    o.device = -1;
    o.num_gpus = 0;   // all visible GPUs; "--gpus N" inside the option string narrows it
    o.cluster_options = search_options.c_str();
    print_message("Running search on the MI355X engine...", 3);
    if (g_verbosity >= 3) putchar('\n');
    // search.rs:45-46 hands <TARGET> to `foldseek search` as the query DB and <INPUT> as the target DB; kept as is
    int rc = uc_search(target.c_str(), input.c_str(), output_aln_db.c_str(), tmp.c_str(), &o, nullptr);   // search.rs:45-55
    if (rc != 0) error(ERR_GENERAL, std::string("search engine failed with code ") + std::to_string(rc) + "\n" + uc_last_error());
    println_message(" Done", 3);
    rc = uc_convertalis(target.c_str(), input.c_str(), output_aln_db.c_str(), output_m8.c_str(), &o);     // search.rs:57-63
    if (rc != 0) error(ERR_GENERAL, std::string("convertalis failed with code ") + std::to_string(rc) + "\n" + uc_last_error());
    if (!keep_aln_db) {                                                                                   // search.rs:66-75
        rc = uc_rmdb(output_aln_db.c_str());
        if (rc != 0) error(ERR_GENERAL, std::string("rmdb failed with code ") + std::to_string(rc) + "\n" + uc_last_error());
    }
    write_checkpoint(parent + "/search.chk", "1");   // search.rs:80
    return 0;
}

// clap's own failures (missing positional, unknown flag, bad value) print "error: ..." + a usage hint on stderr and exit
// with status 2; `arg_required_else_help = true` (arg_parser.rs:8,226) prints the HELP text — also status 2 — when the
// (sub)command gets no argument at all.  ERR_ARGPARSE (0x40) is reserved for cluster.rs:11-15's unwrap_or_else arms, which
// clap's required positionals make unreachable.
constexpr int CLAP_USAGE = 2;
[[noreturn]] void clap_error(bool is_search, const std::string &what) {
    fprintf(stderr, "error: %s\n\nUsage: unicore %s\n\nFor more information, try '--help'.\n", what.c_str(),
            is_search ? "search [OPTIONS] <INPUT> <TARGET> <OUTPUT> <TMP>" : "cluster [OPTIONS] <INPUT> <OUTPUT> <TMP>");
    exit(CLAP_USAGE);
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) { usage(stderr); return CLAP_USAGE; }
    if (!strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) { usage(stdout); return 0; }
    if (!strcmp(argv[1], "version") || !strcmp(argv[1], "--version")) { puts(uc_version()); return 0; }
    const bool is_search = !strcmp(argv[1], "search");
    if (strcmp(argv[1], "cluster") != 0 && !is_search) error(0x30 /* ERR_MODULE_NOT_IMPLEMENTED */, argv[1]);
    if (argc == 2) { if (is_search) usage_search(stderr); else usage(stderr); return CLAP_USAGE; }   // arg_required_else_help
    std::vector<std::string> pos;
    bool keep = false;
    std::string copts = "-c 0.8";   // arg_parser.rs:238-239 (cluster), :262-263 (search)
    int threads = 0, verbosity = 3;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto value = [&]() -> std::string {
            if (i + 1 >= argc) clap_error(is_search, "a value is required for '" + a + "' but none was supplied");
            return argv[++i];
        };
        if (a == "-k" || a == (is_search ? "--keep-aln-db" : "--keep-cluster-db")) keep = true;
        else if (!is_search && (a == "-c" || a == "--cluster-options")) copts = value();
        else if (!is_search && a.rfind("--cluster-options=", 0) == 0) copts = a.substr(18);
        else if (is_search && (a == "-s" || a == "--search-options")) copts = value();
        else if (is_search && a.rfind("--search-options=", 0) == 0) copts = a.substr(17);
        else if (a == "--threads") threads = atoi(value().c_str());
        else if (a == "-v" || a == "--verbosity") verbosity = atoi(value().c_str());
        else if (a == "-h" || a == "--help") { if (is_search) usage_search(stdout); else usage(stdout); return 0; }
        else if (a.size() > 1 && a[0] == '-') clap_error(is_search, "unexpected argument '" + a + "' found");
        else pos.push_back(a);
    }
    const size_t want = is_search ? 4 : 3;
    if (pos.size() < want) clap_error(is_search, "the following required arguments were not provided");
    if (pos.size() > want) clap_error(is_search, "unexpected argument '" + pos[want] + "' found");
    if (verbosity < 0 || verbosity > 4) clap_error(is_search, "invalid value for '--verbosity <VERBOSITY>'");
    g_verbosity = verbosity;
    // set_threads (variables.rs:155-166): 0 -> all CPUs, clamp to the CPU count
    int cpus = (int)std::thread::hardware_concurrency();
    if (cpus < 1) cpus = 1;
    if (threads > cpus) {
        if (g_verbosity >= 2) fprintf(stderr, "Warning: the given number of threads is greater than the number of system CPUs; adjusting to %d\n", cpus);
        threads = cpus;
    }
    if (threads <= 0) threads = cpus;
    const int rc = is_search ? search_run(pos[0], pos[1], pos[2], pos[3], keep, copts, threads) : cluster_run(pos[0], pos[1], pos[2], keep, copts, threads);
    // outputs and the checkpoint are on disk: skip the teardown of the HIP runtime and its parked work buffers (~0.1 s per call)
    fflush(stdout);
    fflush(stderr);
    _exit(rc);
}
