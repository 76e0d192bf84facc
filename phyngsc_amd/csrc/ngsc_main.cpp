// ngsc - command line front end: `ngsc <in.fastq> <out.ngsc> [threads] [--ranks P] [--gpus N | --devices a,b,..]`.
//
// Keeps the reference's positional interface (phyNGSC.cpp:61-108: <input> <output> <threads>, exit codes
// 1 usage, 2 open failure, 3 allocation failure).  `mpiexec -np P` becomes `--ranks P` (virtual ranks,
// default 8): the output is what the reference writes with P processes, up to block order.  The
// <threads> argument is accepted and ignored (the output never depended on it).  `--devices 0,1,2,3` spreads
// the ranks over several GPUs from this one process (P must be a multiple of the device count); the file
// does not depend on how many devices worked on it.  `ngsc -d <in.ngsc> <out.fastq> [--device D]` is the way back
// (ngsc_decompress_file; the reference has no decompressor).
#include "../../include/ngsc.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv)
{
    const char *in = nullptr, *out = nullptr;
    unsigned P = 8;
    int npos = 0;
    bool decode = false;
    std::vector<int> devices;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--ranks") && i + 1 < argc) P = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "-d") || !strcmp(argv[i], "--decompress")) decode = true;
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) {   // the first N devices
            devices.clear();
            for (int d = 0, n = atoi(argv[++i]); d < n; ++d) devices.push_back(d);
        }
        else if ((!strcmp(argv[i], "--device") || !strcmp(argv[i], "--devices")) && i + 1 < argc) {
            devices.clear();
            for (const char *p = argv[++i]; *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') ++p;
                if (*p == ',') ++p;
            }
        }
        else if (npos == 0) { in = argv[i]; ++npos; }
        else if (npos == 1) { out = argv[i]; ++npos; }
        else if (npos == 2) { if (atoi(argv[i]) < 1) { fprintf(stderr, "\n[E] ERROR: Number of threads is less than 1.\n"); return 1; } ++npos; }
        else { npos = 99; }
    }
    if (npos < 2 || npos > 3) {
        fprintf(stderr, "\n[E] ERROR: Incorrect number of arguments. Usage:\n");
        fprintf(stderr, "\n           ngsc input_filename.fastq output_filename.ngsc [num_threads] [--ranks P] [--gpus N | --devices D[,D..]]\n           ngsc -d input_filename.ngsc output_filename.fastq [--device D]\n");
        return 1;
    }
    if (!decode && P < 2) {
        fprintf(stderr, "\n[E] ERROR: Number of ranks (--ranks) is less than 2.\n");
        return 1;
    }
    if (devices.empty()) devices.push_back(0);
    if (!decode && P % devices.size()) {
        fprintf(stderr, "\n[E] ERROR: Number of ranks (%u) is not a multiple of the number of devices (%zu).\n", P, devices.size());
        return 1;
    }
    std::vector<ngsc_ctx *> ctxs;
    auto destroy_all = [&]() { for (ngsc_ctx *c : ctxs) ngsc_ctx_destroy(c); };
    for (int d : devices) {
        ngsc_ctx *ctx = nullptr;
        int rc = ngsc_ctx_create(d, &ctx);
        if (rc) {
            fprintf(stderr, "\n[E] ERROR: no usable HIP device %d (%s). This program has no CPU fallback.\n", d, ngsc_strerror(rc));
            destroy_all();
            return 3;
        }
        ctxs.push_back(ctx);
    }
    double sec[3] = {0, 0, 0};
    if (decode) printf("\n[I] INFO: Decompressing <<%s>> on GPU %d.\n", in, devices[0]);
    else printf("\n[I] INFO: Processing <<%s>> with %u virtual ranks on %zu GPU context(s).\n", in, P, ctxs.size());
    int rc = decode ? ngsc_decompress_file(ctxs[0], in, out, sec) : ngsc_compress_file_multi(ctxs.data(), (uint32_t)ctxs.size(), in, out, P, sec);
    if (rc) {
        fprintf(stderr, "\n[E] ERROR: %s\n", ngsc_strerror(rc));
        if (rc == NGSC_ERR_HIP)
            for (ngsc_ctx *c : ctxs)
                if (*ngsc_last_hip_error(c)) fprintf(stderr, "          device %d: %s\n", ngsc_ctx_device(c), ngsc_last_hip_error(c));
        destroy_all();
        // exit codes of the reference (phyNGSC.cpp:61-108): 1 usage, 2 cannot open, 3 allocation / device failure; the
        // reference has no code for input it cannot compress (it runs into undefined behaviour there): 4 here
        if (rc == NGSC_ERR_IO) return 2;
        if (rc == NGSC_ERR_ARG) return 1;
        return (rc == NGSC_ERR_NOMEM || rc == NGSC_ERR_HIP) ? 3 : 4;
    }
    // (a context's result tables describe its last job only: report the file that was written)
    unsigned long long out_bytes = 0;
    if (FILE *f = fopen(out, "rb")) { fseek(f, 0, SEEK_END); out_bytes = (unsigned long long)ftell(f); fclose(f); }
    if (decode) {
        printf("\nREAD\t\tH2D+DEVICE+D2H\tWRITE\t\tOUT_BYTES\n");
        printf("%f\t%f\t%f\t%llu\n", sec[0], sec[1], sec[2], out_bytes);
    } else {
        printf("\nRANKS\tREAD\t\tH2D+DEVICE+D2H\tWRITE\t\tOUT_BYTES\n");
        printf("%u\t%f\t%f\t%f\t%llu\n", P, sec[0], sec[1], sec[2], out_bytes);
    }
    destroy_all();
    return 0;
}
