/* orc_cli -- the CPU oracle behind Dashing's command lines, ONE PROCESS PER JOB: what bench.py's cpu_baseline times when it runs the
 * sweep "the way DandD drives Dashing" (/root/reference/lib/huffman_dandd.py:214-218: `parallel -j 95% 'dashing sketch -k{} ...'`,
 * lib/sketch_classes.py:358-365, 370-372, 312).  TEST INFRASTRUCTURE like everything under oracle/: never linked or run by the product.
 *   orc_cli sketch [--no-canon] -k<K> -S <P> --prefix <dir> <fasta>     -> <dir>/<basename>.w.<K>.spacing.<P>.hll  (2^P register bytes)
 *   orc_cli union -o <out> <in> ...                                      -> byte max
 *   orc_cli card --presketched <path> ...                                -> "#Path\tSize (est.)" + one line per sketch (lib/sketch_classes.py:320-321)
 * (round 4 ran these jobs as threads of one process: no fork / exec, no file re-open per job -- it flattered the CPU.) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dd_oracle.h"

static unsigned char *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    unsigned char *buf;
    long sz;
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf = (unsigned char *)malloc((size_t)sz + 1);
    if (!buf || fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "read error on %s\n", path); exit(2); }
    fclose(f);
    *n = (size_t)sz;
    return buf;
}
static void spill(const char *path, const unsigned char *buf, size_t n) {
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(buf, 1, n, f) != n || fclose(f) != 0) { perror(path); exit(2); }
}
static int log2_of(size_t n) {
    int p = 0;
    while (((size_t)1 << p) < n) ++p;
    return p;
}

int main(int argc, char **argv) {
    int i;
    if (argc < 2) return 64;
    if (!strcmp(argv[1], "sketch")) {
        int k = 0, p = 0, canon = 1;
        const char *prefix = ".", *fasta = NULL, *base;
        char out[4096];
        size_t n;
        unsigned char *fa, *regs;
        for (i = 2; i < argc; ++i) {
            if (!strncmp(argv[i], "-k", 2)) k = atoi(argv[i] + 2);
            else if (!strcmp(argv[i], "-S") && i + 1 < argc) p = atoi(argv[++i]);
            else if (!strcmp(argv[i], "--prefix") && i + 1 < argc) prefix = argv[++i];
            else if (!strcmp(argv[i], "--no-canon")) canon = 0;
            else fasta = argv[i];
        }
        if (!fasta || k < 1 || k > 64 || p < 4 || p > 20) return 64;
        fa = slurp(fasta, &n);
        regs = (unsigned char *)calloc((size_t)1 << p, 1);
        if (!regs || orc_sketch(fa, n, k, p, canon, regs) != 0) return 1;
        base = strrchr(fasta, '/');
        base = base ? base + 1 : fasta;
        snprintf(out, sizeof out, "%s/%s.w.%d.spacing.%d.hll", prefix, base, k, p);
        spill(out, regs, (size_t)1 << p);
        return 0;
    }
    if (!strcmp(argv[1], "union")) {
        const char *out = NULL;
        unsigned char *acc = NULL;
        size_t m = 0;
        for (i = 2; i < argc; ++i) {
            if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
            else if (!strcmp(argv[i], "-z")) continue;
            else {
                size_t n;
                unsigned char *r = slurp(argv[i], &n);
                if (!acc) acc = r, m = n;
                else {
                    if (n != m) return 1;
                    orc_union(acc, r, m);
                    free(r);
                }
            }
        }
        if (!out || !acc) return 64;
        spill(out, acc, m);
        return 0;
    }
    if (!strcmp(argv[1], "card")) {
        printf("#Path\tSize (est.)\n");
        for (i = 2; i < argc; ++i) {
            size_t n;
            unsigned char *r;
            if (!strcmp(argv[i], "--presketched")) continue;
            r = slurp(argv[i], &n);
            printf("%s\t%.17g\n", argv[i], orc_card(r, log2_of(n)));
            free(r);
        }
        return 0;
    }
    return 64;
}
