// ngsc - command line front end: `ngsc <in.fastq> <out.ngsc> [threads] [--ranks P] [--device D]`.
//
// Keeps the reference's positional interface (phyNGSC.cpp:61-108: <input> <output> <threads>, exit codes
// 1 usage, 2 open failure, 3 allocation failure).  `mpiexec -np P` becomes `--ranks P` (virtual ranks,
// default 8): the output is what the reference writes with P processes, up to block order.  The
// <threads> argument is accepted and ignored (the output never depended on it).
#include "../../include/ngsc.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

int main(int argc, char **argv)
{
    const char *in = nullptr, *out = nullptr;
    unsigned P = 8;
    int device = 0, npos = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--ranks") && i + 1 < argc) P = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (npos == 0) { in = argv[i]; ++npos; }
        else if (npos == 1) { out = argv[i]; ++npos; }
        else if (npos == 2) { if (atoi(argv[i]) < 1) { fprintf(stderr, "\n[E] ERROR: Number of threads is less than 1.\n"); return 1; } ++npos; }
        else { npos = 99; }
    }
    if (npos < 2 || npos > 3) {
        fprintf(stderr, "\n[E] ERROR: Incorrect number of arguments. Usage:\n");
        fprintf(stderr, "\n           ngsc input_filename.fastq output_filename.ngsc [num_threads] [--ranks P] [--device D]\n");
        return 1;
    }
    if (P < 2) {
        fprintf(stderr, "\n[E] ERROR: Number of ranks (--ranks) is less than 2.\n");
        return 1;
    }
    ngsc_ctx *ctx = nullptr;
    int rc = ngsc_ctx_create(device, &ctx);
    if (rc) {
        fprintf(stderr, "\n[E] ERROR: no usable HIP device %d (%s). This program has no CPU fallback.\n", device, ngsc_strerror(rc));
        return 3;
    }
    printf("\n[I] INFO: Processing <<%s>> with %u virtual ranks on GPU %d.\n", in, P, device);
    double sec[3] = {0, 0, 0};
    rc = ngsc_compress_file(ctx, in, out, P, sec);
    if (rc) {
        fprintf(stderr, "\n[E] ERROR: %s%s%s\n", ngsc_strerror(rc), rc == NGSC_ERR_HIP ? ": " : "",
                rc == NGSC_ERR_HIP ? ngsc_last_hip_error(ctx) : "");
        ngsc_ctx_destroy(ctx);
        return rc == NGSC_ERR_IO ? 2 : 3;
    }
    printf("\nRANKS\tREAD+H2D\tDEVICE\t\tD2H+WRITE\tN_SUBBLOCKS\tOUT_BYTES\n");
    printf("%u\t%f\t%f\t%f\t%u\t\t%llu\n", P, sec[0], sec[1], sec[2], ngsc_result_num_subblocks(ctx),
           (unsigned long long)ngsc_result_bytes(ctx));
    ngsc_ctx_destroy(ctx);
    return 0;
}
