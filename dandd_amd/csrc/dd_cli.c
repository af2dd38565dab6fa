/* dd_cli.c -- `dashing`, the reference's REAL plugin API, on the MI355X: a plain C99 program over include/dandd_hip.h
 * (libdandd_hip.so; no Python anywhere) that speaks the three command lines DandD shells out to, so that an UNMODIFIED
 * DandD is pointed at the GPU by putting this executable first on PATH:
 *
 *   dashing sketch [--no-canon] -k<K> -S <R> --prefix <dir> <fasta> ...    /root/reference/lib/sketch_classes.py:351-366
 *        -> <dir>/<basename>.w.<K>.spacing.<R>.hll                          (name: lib/sketch_classes.py:100)
 *   dashing union -z -o <out> <in> ...                                      lib/sketch_classes.py:368-373
 *   dashing card --presketched <path> ...                                   lib/sketch_classes.py:306-316
 *        -> "#Path\tSize (est.)" and one "<path>\t<estimate>" line per sketch on stdout, which is what
 *           lib/sketch_classes.py:318-321 parses
 * (launched K at a time by  parallel -j 95% '<cmd with {}>' ::: k...  , lib/huffman_dandd.py:214-218,233.)
 *
 * Sketch files: this engine's container (8-byte magic, log2m, k, canonical flag, one reserved byte, 2^log2m register bytes;
 * dandd_amd/host/backend.py reads and writes the same), or -- DANDD_SKETCH_FORMAT=dashing | dashing-plain -- Dashing's layout
 * AS RECALLED (unverified against a Dashing binary: DESIGN.md section 6); either is read, gzip'd or not.  `-z` is accepted:
 * the reference never reads sketch bytes itself, and this container is not compressed.
 *
 * A process per command is a hipInit per command.  `dashing serve --socket <path> [--idle-exit <seconds>]` keeps one process
 * with its GPU contexts alive; with DANDD_DASHING_SERVER=<path> in the environment the three commands above are forwarded to it
 * (argv + working directory over a unix socket; exit status, stdout and stderr come back) and run by the same functions.
 * Without a reachable server they run here.  Exit status: 0, 1 on a failed command (message on stderr), 64 on a bad command line.
 *
 * The k-batch.  DandD does not call `dashing sketch` K times: it calls GNU parallel once per node,
 *   parallel -j 95% ' dashing sketch -k{} -S <R> --prefix <dir/k{}> <fasta> ' ::: k1 k2 ...      lib/huffman_dandd.py:214-218,233
 * (and the same around `dashing union`).  Installed under the name `parallel` (dandd_amd/bin/fused/parallel; or `dashing parallel
 * ...`) this program takes that one call as what it is: ONE fused k-sweep over the FASTA (dd_sketch_fasta over every run of
 * consecutive ks: the file read once, all ks in one launch per window class) that leaves the K files the K processes would have
 * left; a `union` template is run k by k in this process.  Any other use -- options other than -j, a command that is not one
 * `dashing sketch | union | card` with plain words -- goes to the next `parallel` on PATH unchanged, or, where there is none, is run
 * value by value through /bin/sh (exit status = failed jobs, as GNU parallel counts them). */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <signal.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/select.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>
#include <zlib.h>

#include "dandd_hip.h"

/* ---- what a command prints: collected, so that a server can send it back -------------------------------------------- */
typedef struct {
    char *p;
    size_t n, cap;
} text;
static void text_add(text *t, const char *fmt, ...) {
    va_list ap;
    int need;
    va_start(ap, fmt);
    need = vsnprintf(NULL, 0, fmt, ap);
    va_end(ap);
    if (need < 0) return;
    if (t->n + (size_t)need + 1 > t->cap) {
        size_t cap = t->cap ? t->cap : 256;
        char *q;
        while (cap < t->n + (size_t)need + 1) cap *= 2;
        q = (char *)realloc(t->p, cap);
        if (!q) return;
        t->p = q;
        t->cap = cap;
    }
    va_start(ap, fmt);
    vsnprintf(t->p + t->n, (size_t)need + 1, fmt, ap);
    va_end(ap);
    t->n += (size_t)need;
}
typedef struct {
    text out, err;
} sink;

/* ---- GPU contexts: one per (log2m, canonical), made on first use, kept for the life of the process ------------------- */
static dd_ctx *g_ctx[21][2];
static dd_ctx *context(int log2m, int canonical, sink *s) {
    dd_ctx **slot;
    if (log2m < 4 || log2m > 20) {
        text_add(&s->err, "dashing: sketch size 2^%d is outside 2^4..2^20\n", log2m);
        return NULL;
    }
    slot = &g_ctx[log2m][canonical ? 1 : 0];
    if (!*slot) {
        const char *dev = getenv("DANDD_DEVICE");
        *slot = dd_create(dev ? atoi(dev) : 0, log2m, canonical ? 1 : 0);
        if (!*slot) text_add(&s->err, "dashing: %s\n", dd_last_error());
    }
    return *slot;
}
static void contexts_destroy(void) {
    int p, c;
    for (p = 0; p <= 20; ++p)
        for (c = 0; c < 2; ++c)
            if (g_ctx[p][c]) dd_destroy(g_ctx[p][c]), g_ctx[p][c] = NULL;
}

/* ---- sketch files --------------------------------------------------------------------------------------------------- */
static const unsigned char kMagic[8] = {'D', 'D', 'H', 'L', 'L', 1, 0, 0};
enum { kNativeHead = 12, kDashHead = 32 };

static int name_k(const char *path) { /* Dashing's container does not hold k: `.w.<k>.spacing.` or `k<k>[nc].hll` */
    const char *base = strrchr(path, '/'), *w;
    base = base ? base + 1 : path;
    w = strstr(base, ".w.");
    if (w && strstr(w, ".spacing.")) return atoi(w + 3);
    for (w = base + strlen(base); w > base; --w)
        if (w[-1] == 'k' && w[0] >= '0' && w[0] <= '9') return atoi(w);
    return 0;
}
/* -> malloc'd registers, or NULL (message in s->err) */
static unsigned char *sketch_read(const char *path, int *log2m, int *k, int *canonical, sink *s) {
    unsigned char head[kDashHead], *regs = NULL;
    gzFile f = gzopen(path, "rb"); /* (reads plain files as they are) */
    int got, p;
    size_t m, have = 0;
    if (!f) {
        text_add(&s->err, "dashing: %s: %s\n", path, strerror(errno));
        return NULL;
    }
    got = gzread(f, head, kNativeHead);
    if (got == kNativeHead && !memcmp(head, kMagic, 8)) {
        p = head[8], *k = head[9], *canonical = head[10] != 0;
    } else {
        uint32_t np;
        size_t n = strlen(path);
        if (got == kNativeHead) got += gzread(f, head + kNativeHead, kDashHead - kNativeHead);
        if (got != kDashHead) goto bad;
        memcpy(&np, head + 20, 4); /* five 32-bit words, then np, then the cached value (a double) */
        p = (int)np, *k = name_k(path);
        *canonical = !(n >= 6 && !strcmp(path + n - 6, "nc.hll")); /* DandD marks non-canonical UNIONS only */
    }
    if (p < 4 || p > 20) goto bad;
    m = (size_t)1 << p;
    regs = (unsigned char *)malloc(m + 1);
    if (!regs) goto bad;
    while (have < m + 1) {
        got = gzread(f, regs + have, (unsigned)(m + 1 - have));
        if (got <= 0) break;
        have += (size_t)got;
    }
    if (have != m) goto bad; /* short, or longer than its header says */
    gzclose(f);
    *log2m = p;
    return regs;
bad:
    gzclose(f);
    free(regs);
    text_add(&s->err, "dashing: %s: neither a dandd_amd nor a Dashing sketch file\n", path);
    return NULL;
}
static int sketch_write(const char *path, const unsigned char *regs, int log2m, int k, int canonical, sink *s) {
    const char *fmt = getenv("DANDD_SKETCH_FORMAT");
    const size_t m = (size_t)1 << log2m;
    unsigned char head[kDashHead];
    size_t nhead;
    char tmp[4200];
    int ok;
    if (snprintf(tmp, sizeof tmp, "%s.%ld.tmp", path, (long)getpid()) >= (int)sizeof tmp) {
        text_add(&s->err, "dashing: %s: name too long\n", path);
        return 1;
    }
    if (!fmt || !strcmp(fmt, "native")) {
        memcpy(head, kMagic, 8);
        head[8] = (unsigned char)log2m, head[9] = (unsigned char)k, head[10] = canonical ? 1 : 0, head[11] = 0;
        nhead = kNativeHead;
    } else if (!strcmp(fmt, "dashing") || !strcmp(fmt, "dashing-plain")) {
        const uint32_t words[6] = {0, 0, 2 /* Ertl MLE */, 3 /* Ertl joint MLE */, 1, (uint32_t)log2m};
        const double value = 0.0;
        memcpy(head, words, 24);
        memcpy(head + 24, &value, 8);
        nhead = kDashHead;
    } else {
        text_add(&s->err, "dashing: DANDD_SKETCH_FORMAT=%s: expected native, dashing or dashing-plain\n", fmt);
        return 1;
    }
    if (fmt && !strcmp(fmt, "dashing")) {
        gzFile f = gzopen(tmp, "wb6");
        ok = f && gzwrite(f, head, (unsigned)nhead) == (int)nhead && gzwrite(f, regs, (unsigned)m) == (int)m;
        if (f && gzclose(f) != Z_OK) ok = 0;
    } else {
        FILE *f = fopen(tmp, "wb");
        ok = f && fwrite(head, 1, nhead, f) == nhead && fwrite(regs, 1, m, f) == m;
        if (f && fclose(f) != 0) ok = 0;
    }
    if (!ok || rename(tmp, path) != 0) { /* (a truncated file must never take a sketch's name: the cache test is "exists and not empty") */
        text_add(&s->err, "dashing: %s: %s\n", path, strerror(errno));
        unlink(tmp);
        return 1;
    }
    return 0;
}

/* ---- the three commands ---------------------------------------------------------------------------------------------- */
static int threads_glued(const char *a) { return a[0] == '-' && a[1] == 'p' && a[2] >= '0' && a[2] <= '9'; }

static int cmd_sketch(int argc, char **argv, sink *s) {
    int k = 0, p = 0, canon = 1, i, npaths = 0, rc = 0;
    const char *prefix = ".";
    char **paths = (char **)calloc((size_t)argc + 1, sizeof *paths);
    dd_ctx *ctx;
    unsigned char *regs;
    if (!paths) return 1;
    for (i = 0; i < argc; ++i) {
        if (!strncmp(argv[i], "-k", 2) && argv[i][2]) k = atoi(argv[i] + 2);
        else if (!strcmp(argv[i], "-k") && i + 1 < argc) k = atoi(argv[++i]);
        else if (!strncmp(argv[i], "-S", 2) && argv[i][2]) p = atoi(argv[i] + 2);
        else if (!strcmp(argv[i], "-S") && i + 1 < argc) p = atoi(argv[++i]);
        else if ((!strcmp(argv[i], "--prefix") || !strcmp(argv[i], "-P")) && i + 1 < argc) prefix = argv[++i];
        else if (!strcmp(argv[i], "--no-canon") || !strcmp(argv[i], "-C")) canon = 0;
        else if (!strcmp(argv[i], "-z")) continue;
        else if (threads_glued(argv[i])) continue; /* (-p<N>, the form lib/sketch_classes.py:361 has commented out) */
        else if ((!strcmp(argv[i], "-p") || !strcmp(argv[i], "--nthreads")) && i + 1 < argc) ++i; /* (threads: the GPU does not care) */
        else if (argv[i][0] == '-' && argv[i][1]) {
            text_add(&s->err, "dashing sketch: unknown option %s\n", argv[i]);
            free(paths);
            return 64;
        } else if (argv[i][0]) paths[npaths++] = argv[i];
    }
    if (!npaths || k < 1 || k > 64 || p < 4 || p > 20) {
        text_add(&s->err, "usage: dashing sketch [--no-canon] -k<1..64> -S <4..20> --prefix <dir> <fasta> ...\n");
        free(paths);
        return 64;
    }
    ctx = context(p, canon, s);
    regs = (unsigned char *)malloc((size_t)npaths << p);
    if (!ctx || !regs) rc = 1;
    /* one file: the text goes through dd_sketch_fasta; several: through the ingestion pipeline, all in one call */
    if (!rc && (npaths == 1 ? dd_sketch_fasta(ctx, paths[0], k, k, regs) : dd_sketch_files(ctx, (const char *const *)paths, npaths, k, k, regs, 0)) != DD_OK) {
        text_add(&s->err, "dashing sketch: %s\n", dd_last_error());
        rc = 1;
    }
    for (i = 0; i < npaths && !rc; ++i) {
        const char *base = strrchr(paths[i], '/');
        char out[4096];
        base = base ? base + 1 : paths[i];
        if (snprintf(out, sizeof out, "%s/%s.w.%d.spacing.%d.hll", prefix, base, k, p) >= (int)sizeof out) {
            text_add(&s->err, "dashing sketch: %s: name too long\n", paths[i]);
            rc = 1;
        } else rc = sketch_write(out, regs + ((size_t)i << p), p, k, canon, s);
    }
    free(regs);
    free(paths);
    return rc;
}

static int cmd_union(int argc, char **argv, sink *s) {
    const char *out = NULL;
    unsigned char **in = (unsigned char **)calloc((size_t)argc + 1, sizeof *in), *merged = NULL;
    int n = 0, i, p = 0, k = 0, canon = 1, rc = 0;
    if (!in) return 1;
    for (i = 0; i < argc && !rc; ++i) {
        if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "-z") || threads_glued(argv[i])) continue;
        else if ((!strcmp(argv[i], "-p") || !strcmp(argv[i], "--nthreads")) && i + 1 < argc) ++i;
        else if (argv[i][0] == '-' && argv[i][1]) {
            text_add(&s->err, "dashing union: unknown option %s\n", argv[i]);
            rc = 64;
        } else {
            int pi, ki, ci;
            in[n] = sketch_read(argv[i], &pi, &ki, &ci, s);
            if (!in[n]) rc = 1;
            else if (n++ == 0) p = pi, k = ki, canon = ci;
            else if (pi != p) {
                text_add(&s->err, "dashing union: %s has 2^%d registers, the first input 2^%d\n", argv[i], pi, p);
                rc = 1;
            }
        }
    }
    if (!rc && (!out || !n)) {
        text_add(&s->err, "usage: dashing union [-z] -o <out> <sketch> ...\n");
        rc = 64;
    }
    if (!rc) {
        dd_ctx *ctx = context(p, canon, s);
        merged = (unsigned char *)malloc((size_t)1 << p);
        if (!ctx || !merged) rc = 1;
        else if (dd_union(ctx, (const uint8_t *const *)in, n, (size_t)1 << p, merged) != DD_OK) {
            text_add(&s->err, "dashing union: %s\n", dd_last_error());
            rc = 1;
        } else {
            const size_t len = strlen(out);
            if (!k) k = name_k(out);
            if (len >= 6 && !strcmp(out + len - 6, "nc.hll")) canon = 0;
            rc = sketch_write(out, merged, p, k, canon, s);
        }
    }
    for (i = 0; i < n; ++i) free(in[i]);
    free(in);
    free(merged);
    return rc;
}

static int cmd_card(int argc, char **argv, sink *s) {
    int i, rc = 0;
    text_add(&s->out, "#Path\tSize (est.)\n"); /* the header lib/sketch_classes.py:320-321 skips */
    for (i = 0; i < argc && !rc; ++i) {
        int p, k, canon;
        unsigned char *regs;
        dd_ctx *ctx;
        double est = 0.0;
        if (!strcmp(argv[i], "--presketched") || !argv[i][0] || threads_glued(argv[i])) continue;
        if ((!strcmp(argv[i], "-p") || !strcmp(argv[i], "--nthreads")) && i + 1 < argc) {
            ++i;
            continue;
        }
        if (argv[i][0] == '-' && argv[i][1]) {
            text_add(&s->err, "dashing card: unknown option %s (only --presketched inputs are supported)\n", argv[i]);
            return 64;
        }
        regs = sketch_read(argv[i], &p, &k, &canon, s);
        if (!regs) return 1;
        ctx = context(p, canon, s);
        if (!ctx || dd_card(ctx, regs, &est) != DD_OK) {
            if (ctx) text_add(&s->err, "dashing card: %s\n", dd_last_error());
            rc = 1;
        } else text_add(&s->out, "%s\t%.17g\n", argv[i], est); /* (round-trips the double through DandD's float()) */
        free(regs);
    }
    return rc;
}

/* ---- `parallel -j N '<dashing ... {} ...>' ::: k ...`: the k-batch of lib/huffman_dandd.py:214-233, fused ------------------------ */
typedef struct {
    char **tok;   /* the command template's words ({} still in them) */
    int ntok;
    char **val;   /* what follows ::: */
    int nval;
    char *buf;    /* the words' storage */
} kbatch;
static void kbatch_free(kbatch *b) {
    free(b->tok);
    free(b->buf);
    memset(b, 0, sizeof *b);
}
/* argv after the program name.  -> 1: a k-batch of `dashing sketch|union|card` this program can run itself (b filled); 0: anything else */
static int kbatch_parse(int argc, char **argv, kbatch *b) {
    int i = 0, n = 0;
    const char *tmpl, *c;
    char *w;
    memset(b, 0, sizeof *b);
    while (i < argc && argv[i][0] == '-' && argv[i][1]) { /* -j N | -jN | --jobs N: how many at once is the GPU's business */
        if ((!strcmp(argv[i], "-j") || !strcmp(argv[i], "--jobs")) && i + 1 < argc) i += 2;
        else if (!strncmp(argv[i], "-j", 2) && argv[i][2]) i += 1;
        else return 0;
    }
    if (i + 2 >= argc || strcmp(argv[i + 1], ":::") != 0) return 0;
    tmpl = argv[i];
    for (c = tmpl; *c; ++c) /* a word is a word only where the shell would leave it alone */
        if (strchr("|&;<>()$`\\\"'*?[]#~", *c)) return 0;
    b->val = argv + i + 2, b->nval = argc - i - 2;
    for (i = 0; i < b->nval; ++i) { /* ks: positive integers */
        const char *v = b->val[i];
        if (!*v || strlen(v) > 3) return 0;
        for (c = v; *c; ++c)
            if (*c < '0' || *c > '9') return 0;
        if (atoi(v) < 1) return 0;
    }
    b->buf = (char *)malloc(strlen(tmpl) + 1);
    b->tok = (char **)calloc(strlen(tmpl) / 2 + 2, sizeof *b->tok);
    if (!b->buf || !b->tok) {
        kbatch_free(b);
        return 0;
    }
    strcpy(b->buf, tmpl);
    for (w = strtok(b->buf, " \t\n"); w; w = strtok(NULL, " \t\n")) b->tok[n++] = w;
    b->ntok = n;
    if (n >= 2) {
        const char *base = strrchr(b->tok[0], '/');
        base = base ? base + 1 : b->tok[0];
        if (!strcmp(base, "dashing") && (!strcmp(b->tok[1], "sketch") || !strcmp(b->tok[1], "union") || !strcmp(b->tok[1], "card"))) return 1;
    }
    kbatch_free(b);
    return 0;
}
/* word with every {} replaced by v, malloc'd */
static char *subst(const char *word, const char *v) {
    size_t n = 1, lv = strlen(v);
    const char *c;
    char *out, *o;
    for (c = word; *c; ++c) n += (c[0] == '{' && c[1] == '}') ? lv : 1;
    out = o = (char *)malloc(n + 1);
    if (!out) return NULL;
    for (c = word; *c;) {
        if (c[0] == '{' && c[1] == '}') memcpy(o, v, lv), o += lv, c += 2;
        else *o++ = *c++;
    }
    *o = 0;
    return out;
}
static int by_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

static int run_command(int argc, char **argv, sink *s);
static int cmd_parallel(int argc, char **argv, sink *s) {
    kbatch b;
    int rc = 0, i, j;
    if (!kbatch_parse(argc, argv, &b)) {
        text_add(&s->err, "dashing parallel: only  [-j N] '<dashing sketch|union|card ... {} ...>' ::: k ...  is run here\n");
        return 64;
    }
    if (!strcmp(b.tok[1], "sketch")) {
        /* the fused form: k only in -k{} and in the prefix; one FASTA (what DandD passes) or several */
        int p = 0, canon = 1, npaths = 0, fusable = 1, *ks = (int *)malloc((size_t)b.nval * sizeof *ks);
        const char *prefix = ".";
        char **paths = (char **)calloc((size_t)b.ntok + 1, sizeof *paths);
        if (!ks || !paths) rc = 1;
        for (i = 2; i < b.ntok && !rc; ++i) {
            const char *t = b.tok[i];
            if (!strcmp(t, "-k{}")) continue;
            if (!strcmp(t, "-k") && i + 1 < b.ntok && !strcmp(b.tok[i + 1], "{}")) ++i;
            else if (!strncmp(t, "-S", 2) && t[2]) p = atoi(t + 2);
            else if (!strcmp(t, "-S") && i + 1 < b.ntok) p = atoi(b.tok[++i]);
            else if ((!strcmp(t, "--prefix") || !strcmp(t, "-P")) && i + 1 < b.ntok) prefix = b.tok[++i];
            else if (!strcmp(t, "--no-canon") || !strcmp(t, "-C")) canon = 0;
            else if (!strcmp(t, "-z") || threads_glued(t)) continue;
            else if ((!strcmp(t, "-p") || !strcmp(t, "--nthreads")) && i + 1 < b.ntok) ++i;
            else if (t[0] == '-' || strstr(t, "{}")) fusable = 0; /* (k somewhere this form does not know: each k through cmd_sketch below) */
            else paths[npaths++] = b.tok[i];
        }
        if (!rc && fusable && npaths && p >= 4 && p <= 20) {
            dd_ctx *ctx = context(p, canon, s);
            const size_t m = (size_t)1 << p;
            for (i = 0; i < b.nval; ++i) ks[i] = atoi(b.val[i]);
            qsort(ks, (size_t)b.nval, sizeof *ks, by_int);
            if (!ctx) rc = 1;
            for (i = 0; i < b.nval && !rc;) { /* every run of consecutive ks: one sweep */
                int hi = i, K, q;
                unsigned char *regs;
                while (hi + 1 < b.nval && ks[hi + 1] <= ks[hi] + 1) ++hi;
                K = ks[hi] - ks[i] + 1;
                if (ks[i] < 1 || ks[hi] > 64) {
                    text_add(&s->err, "dashing parallel: k %d..%d is outside 1..64\n", ks[i], ks[hi]);
                    rc = 1;
                    break;
                }
                regs = (unsigned char *)malloc((size_t)npaths * (size_t)K * m);
                if (!regs) rc = 1;
                else if ((npaths == 1 ? dd_sketch_fasta(ctx, paths[0], ks[i], ks[hi], regs)
                                      : dd_sketch_files(ctx, (const char *const *)paths, npaths, ks[i], ks[hi], regs, 0)) != DD_OK) {
                    text_add(&s->err, "dashing parallel: %s\n", dd_last_error());
                    rc = 1;
                }
                for (q = 0; q < npaths && !rc; ++q)
                    for (j = 0; j < K && !rc; ++j) {
                        char kstr[8], out[4096], *dir;
                        const char *base = strrchr(paths[q], '/');
                        base = base ? base + 1 : paths[q];
                        snprintf(kstr, sizeof kstr, "%d", ks[i] + j);
                        dir = subst(prefix, kstr);
                        if (!dir || snprintf(out, sizeof out, "%s/%s.w.%d.spacing.%d.hll", dir, base, ks[i] + j, p) >= (int)sizeof out) {
                            text_add(&s->err, "dashing parallel: %s: name too long\n", paths[q]);
                            rc = 1;
                        } else rc = sketch_write(out, regs + ((size_t)q * (size_t)K + (size_t)j) * m, p, ks[i] + j, canon, s);
                        free(dir);
                    }
                free(regs);
                i = hi + 1;
            }
            free(ks);
            free(paths);
            kbatch_free(&b);
            return rc;
        }
        free(ks);
        free(paths);
        if (rc) {
            kbatch_free(&b);
            return rc;
        }
    }
    /* union, card, and sketch templates the fused form does not cover: value by value, in this process */
    for (i = 0; i < b.nval; ++i) {
        char **av = (char **)calloc((size_t)b.ntok + 1, sizeof *av);
        int one = av ? 0 : 1;
        for (j = 1; j < b.ntok && !one; ++j)
            if (!(av[j - 1] = subst(b.tok[j], b.val[i]))) one = 1;
        if (!one) one = run_command(b.ntok - 1, av, s);
        for (j = 0; av && j < b.ntok; ++j) free(av[j]);
        free(av);
        if (one) ++rc; /* (GNU parallel's exit status: the number of jobs that failed) */
    }
    kbatch_free(&b);
    return rc > 101 ? 101 : rc;
}

static int run_command(int argc, char **argv, sink *s) {
    if (argc >= 1 && !strcmp(argv[0], "parallel")) return cmd_parallel(argc - 1, argv + 1, s);
    if (argc >= 1 && !strcmp(argv[0], "sketch")) return cmd_sketch(argc - 1, argv + 1, s);
    if (argc >= 1 && !strcmp(argv[0], "union")) return cmd_union(argc - 1, argv + 1, s);
    if (argc >= 1 && !strcmp(argv[0], "card")) return cmd_card(argc - 1, argv + 1, s);
    text_add(&s->err, "dashing (dandd_amd, MI355X): sketch | union | card | serve -- the commands DandD issues (ABI %d)\n", dd_abi_version());
    return 64;
}

/* ---- the resident form -------------------------------------------------------------------------------------------------
 * request : u32 argc, then argc + 1 strings (u32 length + bytes): the arguments and the client's working directory
 * reply   : i32 exit status, u32 + stdout bytes, u32 + stderr bytes
 * argv[0] = "ping": "are you there" (status 0, stdout = the server's pid); argv[0] = "shutdown": the server leaves */
static int io_all(int fd, void *buf, size_t n, int writing) {
    char *p = (char *)buf;
    while (n) {
        const ssize_t r = writing ? write(fd, p, n) : read(fd, p, n);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return -1;
        p += r, n -= (size_t)r;
    }
    return 0;
}
static int put_blob(int fd, const char *p, size_t n) {
    uint32_t len = (uint32_t)n;
    return io_all(fd, &len, 4, 1) || (n && io_all(fd, (void *)p, n, 1)) ? -1 : 0;
}
static char *get_blob(int fd, uint32_t limit) {
    uint32_t len;
    char *p;
    if (io_all(fd, &len, 4, 0) || len > limit || !(p = (char *)malloc((size_t)len + 1))) return NULL;
    if (len && io_all(fd, p, len, 0)) {
        free(p);
        return NULL;
    }
    p[len] = 0;
    return p;
}
static int unix_socket(const char *path, struct sockaddr_un *sa) {
    int fd;
    if (strlen(path) >= sizeof sa->sun_path) return -1;
    memset(sa, 0, sizeof *sa);
    sa->sun_family = AF_UNIX;
    strcpy(sa->sun_path, path);
    fd = socket(AF_UNIX, SOCK_STREAM, 0);
    return fd;
}

static int serve(int argc, char **argv) {
    const char *path = NULL;
    double idle = 0.0;
    struct sockaddr_un sa;
    char home[4096];
    int i, srv, served = 0, leave = 0;
    mode_t old;
    for (i = 0; i < argc; ++i) {
        if (!strcmp(argv[i], "--socket") && i + 1 < argc) path = argv[++i];
        else if (!strcmp(argv[i], "--idle-exit") && i + 1 < argc) idle = atof(argv[++i]);
    }
    if (!path || !getcwd(home, sizeof home)) {
        fprintf(stderr, "usage: dashing serve --socket <path> [--idle-exit <seconds>]\n");
        return 64;
    }
    unlink(path);
    srv = unix_socket(path, &sa);
    old = umask(0177); /* the socket runs commands as this user: nobody else may connect, from the moment it exists */
    if (srv < 0 || bind(srv, (struct sockaddr *)&sa, sizeof sa) != 0 || listen(srv, 64) != 0) {
        perror(path);
        umask(old);
        return 1;
    }
    umask(old);
    signal(SIGPIPE, SIG_IGN); /* a client that hangs up before its reply is written costs a failed write(), not the server */
    printf("dashing serve: listening on %s\n", path);
    fflush(stdout);
    while (!leave) {
        int conn;
        uint32_t n, j;
        char **av = NULL, *cwd = NULL;
        sink s;
        int32_t rc = 64;
        if (idle > 0) {
            fd_set fds;
            struct timeval tv;
            FD_ZERO(&fds);
            FD_SET(srv, &fds);
            tv.tv_sec = (long)idle, tv.tv_usec = (long)((idle - (double)(long)idle) * 1e6);
            if (select(srv + 1, &fds, NULL, NULL, &tv) == 0) break;
        }
        conn = accept(srv, NULL, NULL);
        if (conn < 0) {
            if (errno == EINTR) continue;
            break;
        }
        memset(&s, 0, sizeof s);
        {   /* ... and one that connects and then says nothing holds the (single) line for ten seconds, not for ever */
            struct timeval patience;
            patience.tv_sec = 10, patience.tv_usec = 0;
            (void)setsockopt(conn, SOL_SOCKET, SO_RCVTIMEO, &patience, sizeof patience);
            (void)setsockopt(conn, SOL_SOCKET, SO_SNDTIMEO, &patience, sizeof patience);
        }
        /* a client that goes away or sends nonsense costs its own connection, never the server (its contexts are the point) */
        if (io_all(conn, &n, 4, 0) == 0 && n <= 65536 && (av = (char **)calloc((size_t)n + 1, sizeof *av)) != NULL) {
            for (j = 0; j < n && (av[j] = get_blob(conn, 1u << 20)) != NULL; ++j) {}
            if (j == n && (cwd = get_blob(conn, 1u << 16)) != NULL) {
                if (n == 0 || !strcmp(av[0], "ping")) rc = 0, text_add(&s.out, "%ld\n", (long)getpid());
                else if (!strcmp(av[0], "shutdown")) rc = 0, leave = 1, text_add(&s.out, "%d\n", served);
                else if (chdir(cwd) != 0) rc = 1, text_add(&s.err, "dashing serve: %s: %s\n", cwd, strerror(errno));
                else {
                    rc = run_command((int)n, av, &s);
                    ++served;
                    if (chdir(home) != 0) leave = 1;
                }
                if (io_all(conn, &rc, 4, 1) == 0 && put_blob(conn, s.out.p, s.out.n) == 0) (void)put_blob(conn, s.err.p, s.err.n);
            }
            for (j = 0; j < n; ++j) free(av[j]);
        }
        free(av);
        free(cwd);
        free(s.out.p);
        free(s.err.p);
        close(conn);
    }
    close(srv);
    unlink(path);
    contexts_destroy();
    return 0;
}

/* -> exit status from the server, or -1 when nobody answers at `path` (the caller runs the command itself) */
static int forward(const char *path, int argc, char **argv) {
    struct sockaddr_un sa;
    char cwd[4096], *out = NULL, *err = NULL;
    uint32_t n = (uint32_t)argc;
    int32_t rc = -1;
    int fd = unix_socket(path, &sa), i, ok;
    if (fd < 0 || !getcwd(cwd, sizeof cwd) || connect(fd, (struct sockaddr *)&sa, sizeof sa) != 0) {
        if (fd >= 0) close(fd);
        return -1;
    }
    ok = io_all(fd, &n, 4, 1) == 0;
    for (i = 0; ok && i < argc; ++i) ok = put_blob(fd, argv[i], strlen(argv[i])) == 0;
    ok = ok && put_blob(fd, cwd, strlen(cwd)) == 0 && io_all(fd, &rc, 4, 0) == 0 && (out = get_blob(fd, 1u << 30)) != NULL &&
         (err = get_blob(fd, 1u << 30)) != NULL;
    close(fd);
    if (ok) {
        fputs(out, stdout);
        fputs(err, stderr);
    }
    free(out);
    free(err);
    return ok ? (int)rc : -1;
}

/* ---- under the name `parallel`: what is not the k-batch goes to the real one ---------------------------------------------- */
static int named_parallel(const char *argv0) {
    const char *base = strrchr(argv0, '/');
    return !strcmp(base ? base + 1 : argv0, "parallel");
}
/* the next `parallel` on PATH that is not this program (malloc'd), or NULL */
static char *next_parallel(void) {
    const char *path = getenv("PATH"), *c;
    char self[4096], cand[4096], real[4096];
    ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    if (!path) return NULL;
    self[n > 0 ? n : 0] = 0;
    for (c = path; *c;) {
        const char *e = strchr(c, ':');
        const size_t len = e ? (size_t)(e - c) : strlen(c);
        if (len && len + 10 < sizeof cand) {
            memcpy(cand, c, len);
            strcpy(cand + len, "/parallel");
            if (access(cand, X_OK) == 0 && realpath(cand, real) && strcmp(real, self) != 0) {
                char *out = (char *)malloc(strlen(cand) + 1);
                if (out) strcpy(out, cand);
                return out;
            }
        }
        if (!e) break;
        c = e + 1;
    }
    return NULL;
}
/* (nothing in this process has touched the GPU yet: replacing it is safe) */
static int other_parallel(int argc, char **argv) {
    char *real = next_parallel();
    int i = 0, failed = 0;
    if (real) {
        char **av = (char **)calloc((size_t)argc + 2, sizeof *av);
        if (!av) return 255;
        av[0] = real;
        memcpy(av + 1, argv, (size_t)argc * sizeof *av);
        execv(real, av);
        perror(real);
        return 127;
    }
    /* no GNU parallel here: its simplest form, one job at a time */
    while (i < argc && argv[i][0] == '-' && argv[i][1]) {
        if ((!strcmp(argv[i], "-j") || !strcmp(argv[i], "--jobs")) && i + 1 < argc) i += 2;
        else if (!strncmp(argv[i], "-j", 2) && argv[i][2]) i += 1;
        else break;
    }
    if (i + 2 >= argc || strcmp(argv[i + 1], ":::") != 0) {
        fprintf(stderr, "parallel (dandd_amd): only  [-j N] '<command with {}>' ::: value ...  is understood here, and no other `parallel` is on PATH\n");
        return 255;
    }
    {
        const char *tmpl = argv[i];
        for (i += 2; i < argc; ++i) {
            char *cmd = subst(tmpl, argv[i]);
            const int st = cmd ? system(cmd) : -1;
            free(cmd);
            if (st != 0) ++failed;
        }
    }
    return failed > 101 ? 101 : failed;
}

int main(int argc, char **argv) {
    const char *server = getenv("DANDD_DASHING_SERVER");
    sink s;
    int rc, nc = argc - 1;
    char **cv = argv + 1, **owned = NULL;
    if (argc >= 2 && !strcmp(argv[1], "serve") && !named_parallel(argv[0])) return serve(argc - 2, argv + 2);
    if (named_parallel(argv[0]) || (argc >= 2 && !strcmp(argv[1], "parallel"))) {
        const int skip = named_parallel(argv[0]) ? 1 : 2;
        kbatch b;
        if (!kbatch_parse(argc - skip, argv + skip, &b)) return other_parallel(argc - skip, argv + skip);
        kbatch_free(&b);
        if (skip == 1) { /* the command as the server and run_command know it: "parallel" first */
            owned = (char **)calloc((size_t)argc + 1, sizeof *owned);
            if (!owned) return 1;
            owned[0] = (char *)"parallel";
            memcpy(owned + 1, argv + 1, (size_t)(argc - 1) * sizeof *owned);
            cv = owned, nc = argc;
        }
    }
    if (server && *server && nc >= 1) {
        rc = forward(server, nc, cv);
        if (rc >= 0) {
            free(owned);
            return rc;
        }
        if (getenv("DANDD_SERVER_REQUIRED") && !strcmp(getenv("DANDD_SERVER_REQUIRED"), "1")) {
            fprintf(stderr, "dashing: no server at %s\n", server);
            free(owned);
            return 111;
        }
    }
    memset(&s, 0, sizeof s);
    rc = run_command(nc, cv, &s);
    free(owned);
    if (s.out.p) fputs(s.out.p, stdout);
    if (s.err.p) fputs(s.err.p, stderr);
    fflush(stdout);
    free(s.out.p);
    free(s.err.p);
    contexts_destroy();
    return rc;
}
