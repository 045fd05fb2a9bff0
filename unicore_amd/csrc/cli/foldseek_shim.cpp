// foldseek_shim.cpp — `foldseek`-argv-compatible executable for the sub-commands the reference's cluster
// path spawns, so that an UNMODIFIED Rust `unicore` can be pointed at this engine through path.cfg
// (`foldseek=<path>`, /root/reference/path.cfg:2; accepted by `unicore config --set-foldseek` because
// `<bin> version` exits 0, src/modules/config.rs:49-66).
//   cluster   --threads T -v V <db> <out>_cluster <tmp> <opts...>     cluster.rs:45-49
//   createtsv --threads T -v V <db> <db> <out>_cluster <out>.tsv      cluster.rs:59-62
//   rmdb      <out>_cluster -v V                                       cluster.rs:67-73
//   search    --threads T <queryDB> <targetDB> <out>_aln <tmp> <opts...>   search.rs:44-50
//   convertalis --threads T <queryDB> <targetDB> <out>_aln <out>.m8        search.rs:57-60
//   createdb  <fasta> <db> --prostt5-model <dir> [--gpu 1] --threads T     createdb.rs:157-166 (ProstT5 AA -> 3Di on the GPU)
//   version
// Exit status is the only error channel (src/util/command.rs:10-14): 0 ok, non-zero + stderr text otherwise.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <string>
#include <vector>

#include "unicore_cluster.h"

// A one-shot process owes nobody a tidy teardown: every output file is closed by the time a command returns, and unwinding the HIP
// runtime with tens of GB of parked work buffers costs ~0.1 s of the ~1.2 s an unmodified Unicore waits per `foldseek cluster` spawn
// (tools/cold_stamps.sh).  Successful commands therefore leave through _exit after flushing the standard streams.
[[noreturn]] static void leave(int rc) {
    fflush(stdout);
    fflush(stderr);
    _exit(rc);
}

static int die(int rc) {
    fprintf(stderr, "Error: %s\n", uc_last_error());
    return rc ? rc : 1;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: foldseek <cluster|createtsv|search|convertalis|createdb|databases|rmdb|version> ...\n"); return 2; }
    const std::string cmd = argv[1];
    if (cmd == "version") { puts(uc_version()); return 0; }
    // Flags may appear anywhere (the reference puts them after the positionals, cluster.rs:45-49, but
    // `foldseek cluster -c 0.8 db out tmp` is just as valid).  Whether a flag takes a value is decided by the engine's
    // flag table (uc_option_arity), never by position; --threads / -v are consumed here, the other flags are forwarded
    // verbatim as the option string and validated by the engine (unknown flags are rejected there, not ignored).
    std::vector<std::string> pos;
    std::string opts;
    int threads = 1, verbosity = 3;
    auto is_flag = [](const char *a) { return a[0] == '-' && a[1] != 0 && !((a[1] >= '0' && a[1] <= '9') || a[1] == '.'); };
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (!is_flag(argv[i])) { pos.push_back(a); continue; }
        const int ar = uc_option_arity(a.c_str());
        std::string val;
        bool has_val = false;
        if (ar == 1) {
            if (i + 1 >= argc) { fprintf(stderr, "Error: option %s needs a value\n", a.c_str()); return 2; }
            val = argv[++i]; has_val = true;
        } else if (ar == 2 && i + 1 < argc && (!strcmp(argv[i + 1], "0") || !strcmp(argv[i + 1], "1"))) {
            val = argv[++i]; has_val = true;
        }
        if (a == "--threads") threads = atoi(val.c_str());
        else if (a == "-v") verbosity = atoi(val.c_str());
        else opts += (opts.empty() ? "" : " ") + a + (has_val ? " " + val : "");
    }
    if (cmd == "createdb") {
        // createdb <fasta> [<fasta> ...] <db> --prostt5-model <dir> [--gpu 0|1] [--gpus N] [--threads T] [-v V]        createdb.rs:157-166
        std::vector<std::string> cpos;
        std::string model;
        int cverb = 3, cgpus = 0;      // all visible GPUs: one encoder replica each, sequences sharded over them (`--gpus N` narrows it, as for cluster)
        for (int i = 2; i < argc; i++) {
            const std::string a = argv[i];
            if ((a == "--prostt5-model" || a == "--gpu" || a == "--gpus" || a == "--threads" || a == "-v") && i + 1 < argc) {
                const char *v = argv[++i];
                if (a == "--prostt5-model") model = v;
                else if (a == "-v") cverb = atoi(v);
                else if (a == "--gpus") cgpus = atoi(v);
            } else if (a.size() > 1 && a[0] == '-') { fprintf(stderr, "Error: createdb: option %s is not provided by this engine\n", a.c_str()); return 2; }
            else cpos.push_back(a);
        }
        if (cpos.size() < 2 || model.empty()) { fprintf(stderr, "Error: createdb expects <fasta> [...] <db> --prostt5-model <dir> (structure input is not provided by this engine)\n"); return 2; }
        std::vector<const char *> fp;
        for (size_t i = 0; i + 1 < cpos.size(); i++) fp.push_back(cpos[i].c_str());
        uc_opts co;
        memset(&co, 0, sizeof co);
        co.struct_size = sizeof co; co.threads = 1; co.verbosity = cverb; co.device = -1; co.num_gpus = cgpus;
        int rc = uc_createdb(fp.data(), (int)fp.size(), cpos.back().c_str(), model.c_str(), &co, nullptr);
        if (rc) return die(rc);
        leave(0);
    }
    uc_opts o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    o.threads = threads > 0 ? threads : 1;
    o.verbosity = verbosity;
    o.device = -1;
    o.num_gpus = 0;   // all visible GPUs (north star); `--gpus N` in the options narrows it
    o.cluster_options = opts.c_str();
    if (cmd == "cluster") {
        if (pos.size() != 3) { fprintf(stderr, "Error: cluster expects <db> <out_cluster_db> <tmp>\n"); return 2; }
        int rc = uc_cluster(pos[0].c_str(), pos[1].c_str(), pos[2].c_str(), &o, nullptr);
        if (rc) return die(rc);
        leave(0);
    }
    if (cmd == "createtsv") {
        if (pos.size() != 4) { fprintf(stderr, "Error: createtsv expects <db> <db> <cluster_db> <out.tsv>\n"); return 2; }
        int rc = uc_createtsv(pos[0].c_str(), pos[2].c_str(), pos[3].c_str(), &o);
        if (rc) return die(rc);
        leave(0);
    }
    if (cmd == "search") {
        if (pos.size() != 4) { fprintf(stderr, "Error: search expects <queryDB> <targetDB> <alnDB> <tmp>\n"); return 2; }
        int rc = uc_search(pos[0].c_str(), pos[1].c_str(), pos[2].c_str(), pos[3].c_str(), &o, nullptr);
        if (rc) return die(rc);
        leave(0);
    }
    if (cmd == "convertalis") {
        if (pos.size() != 4) { fprintf(stderr, "Error: convertalis expects <queryDB> <targetDB> <alnDB> <out.m8>\n"); return 2; }
        int rc = uc_convertalis(pos[0].c_str(), pos[1].c_str(), pos[2].c_str(), pos[3].c_str(), &o);
        if (rc) return die(rc);
        leave(0);
    }
    if (cmd == "databases") {
        // `foldseek databases ProstT5 <model> <tmp> --threads T` (createdb.rs:149-155): Unicore calls it when <model>/prostt5-f16.gguf is
        // missing, to DOWNLOAD the weights.  This engine has no network code: it answers the call with where the file must go.
        if (pos.size() < 2) { fprintf(stderr, "Error: databases expects <name> <outDir> <tmp>\n"); return 2; }
        if (pos[0] != "ProstT5") { fprintf(stderr, "Error: database '%s' is not known to this engine (only ProstT5 is ever requested by unicore, createdb.rs:152)\n", pos[0].c_str()); return 2; }
        const std::string f = pos[1] + "/prostt5-f16.gguf";
        if (FILE *fp = fopen(f.c_str(), "rb")) { fclose(fp); printf("ProstT5 weights already in place: %s\n", f.c_str()); return 0; }
        fprintf(stderr, "Error: this engine does not download model weights.  Place the ProstT5 encoder + 3Di head as\n  %s\n"
                        "(GGUF, F16 / F32 tensors: what `foldseek databases ProstT5 <dir> <tmp>` of a stock Foldseek >= 10 fetches) and run `unicore createdb` again.\n", f.c_str());
        return 1;
    }
    if (cmd == "rmdb") {
        if (pos.size() != 1) { fprintf(stderr, "Error: rmdb expects <db>\n"); return 2; }
        int rc = uc_rmdb(pos[0].c_str());
        if (rc) return die(rc);
        leave(0);
    }
    fprintf(stderr, "Error: sub-command '%s' is not provided by this engine (cluster, createtsv, search, convertalis, createdb, databases, rmdb, version)\n", cmd.c_str());
    return 2;
}
