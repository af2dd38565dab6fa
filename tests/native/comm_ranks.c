/* One rank of a multi-GPU job written against include/dandd_hip.h ALONE -- plain C99, no Python, no torch: what a cgo / JNI /
 * FFI host binding of the boundary gets.  Every rank sketches its own synthetic genome on its own GPU
 * (device = rank modulo the GPUs visible), the roots meet in dd_allreduce_max_u8 (RCCL: ncclAllReduce, ncclUint8, ncclMax) and
 * the leaf slabs in dd_allgather_u8; both are checked against what this rank computes alone for every rank's genome.
 *     comm_ranks RANK WORLD IDFILE      (rank 0 writes the communicator's 128-byte id to IDFILE, the others wait for it)
 * exit 0 = ok; 77 = RCCL refused the communicator (two ranks on one GPU: "Duplicate GPU"); anything else = wrong. */
#define __HIP_PLATFORM_AMD__ 1
#define _DEFAULT_SOURCE 1 /* usleep */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "dandd_hip.h"

enum { LOG2M = 12, KMIN = 15, KMAX = 17, K = KMAX - KMIN + 1, NBASES = 60000 };

static size_t genome_of(int rank, unsigned char *fa) { /* ">g<rank>\n" + NBASES bases from a 64-bit LCG, 70 to a line */
    unsigned long long s = 0x9E3779B97F4A7C15ull * (unsigned long long)(rank + 1);
    size_t n = (size_t)sprintf((char *)fa, ">g%d\n", rank);
    int i;
    for (i = 0; i < NBASES; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        fa[n++] = (unsigned char)"ACGT"[(s >> 33) & 3u];
        if (i % 70 == 69) fa[n++] = '\n';
    }
    fa[n++] = '\n';
    return n;
}

int main(int argc, char **argv) {
    int rank, world, ndev = 0, r, info_rank = -1, info_world = -1;
    unsigned long long nred = 0, ngat = 0;
    const size_t m = (size_t)1 << LOG2M, slab = (size_t)K * m;
    unsigned char id[DD_COMM_ID_BYTES], *fa, *mine, *theirs, *want_root, *got;
    unsigned char *dev_root = NULL, *dev_all = NULL;
    dd_ctx *ctx;
    FILE *f;
    size_t n, i;
    if (argc != 4) return 2;
    rank = atoi(argv[1]);
    world = atoi(argv[2]);
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return 3;
    ctx = dd_create(rank % ndev, LOG2M, 1);
    if (!ctx) { fprintf(stderr, "dd_create: %s\n", dd_last_error()); return 4; }
    if (rank == 0) {
        char tmp[4096];
        if (dd_comm_unique_id(id) != DD_OK) { fprintf(stderr, "dd_comm_unique_id: %s\n", dd_last_error()); return 5; }
        snprintf(tmp, sizeof tmp, "%s.tmp", argv[3]);
        f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 || rename(tmp, argv[3]) != 0) return 6;
    } else {
        for (r = 0; r < 600 && access(argv[3], R_OK) != 0; ++r) usleep(100000);
        f = fopen(argv[3], "rb");
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) return 7;
        fclose(f);
    }
    if (dd_comm_init(ctx, rank, world, id) != DD_OK) {
        fprintf(stderr, "dd_comm_init: %s\n", dd_last_error());
        dd_destroy(ctx);
        return 77;
    }
    if (dd_comm_init(ctx, rank, world, id) == DD_OK) return 8; /* a context belongs to one communicator at a time */
    fa = (unsigned char *)malloc(NBASES + NBASES / 70 + 64);
    mine = (unsigned char *)malloc(slab);
    theirs = (unsigned char *)malloc(slab);
    want_root = (unsigned char *)calloc(slab, 1);
    got = (unsigned char *)malloc(slab * (size_t)world);
    if (!fa || !mine || !theirs || !want_root || !got) return 9;
    n = genome_of(rank, fa);
    if (dd_sketch_buffer(ctx, fa, n, KMIN, KMAX, mine) != DD_OK) { fprintf(stderr, "sketch: %s\n", dd_last_error()); return 10; }
    if (hipSetDevice(rank % ndev) != hipSuccess || hipMalloc((void **)&dev_root, slab) != hipSuccess ||
        hipMalloc((void **)&dev_all, slab * (size_t)world) != hipSuccess) return 11;
    if (hipMemcpy(dev_root, mine, slab, hipMemcpyHostToDevice) != hipSuccess) return 12;
    /* the two exchanges of the N > 1 path */
    if (dd_allgather_u8(ctx, dev_root, slab, dev_all) != DD_OK) { fprintf(stderr, "allgather: %s\n", dd_last_error()); return 13; }
    if (dd_allreduce_max_u8(ctx, dev_root, slab) != DD_OK) { fprintf(stderr, "allreduce: %s\n", dd_last_error()); return 14; }
    if (dd_synchronize(ctx) != DD_OK) return 15;
    if (hipMemcpy(got, dev_all, slab * (size_t)world, hipMemcpyDeviceToHost) != hipSuccess) return 16;
    for (r = 0; r < world; ++r) { /* what every rank must have sent, computed here alone */
        n = genome_of(r, fa);
        if (dd_sketch_buffer(ctx, fa, n, KMIN, KMAX, theirs) != DD_OK) return 17;
        if (memcmp(got + (size_t)r * slab, theirs, slab) != 0) { fprintf(stderr, "rank %d: all-gather slot %d differs\n", rank, r); return 18; }
        for (i = 0; i < slab; ++i)
            if (theirs[i] > want_root[i]) want_root[i] = theirs[i];
    }
    if (hipMemcpy(got, dev_root, slab, hipMemcpyDeviceToHost) != hipSuccess) return 19;
    if (memcmp(got, want_root, slab) != 0) { fprintf(stderr, "rank %d: all-reduced root differs from the byte-max of all ranks' sketches\n", rank); return 20; }
    if (dd_comm_info(ctx, &info_rank, &info_world, &nred, &ngat) != DD_OK || info_rank != rank || info_world != world || nred != 1 || ngat != 1) return 21;
    if (dd_comm_destroy(ctx) != DD_OK) return 22;
    if (dd_comm_info(ctx, NULL, &info_world, NULL, NULL) != DD_OK || info_world != 0) return 23;
    (void)hipFree(dev_root);
    (void)hipFree(dev_all);
    dd_destroy(ctx);
    printf("comm_ranks: ok, rank %d of %d on device %d\n", rank, world, rank % ndev);
    return 0;
}
