"""Independent pure-Python statements of the published algorithms (Wang mix, HLL register rule,
FASTA tokenisation, exact distinct canonical k-mers, Ertl MLE).  Used ONLY to pin the C oracle on
small cases -- arbitrary-precision Python ints, no ctypes, nothing shared with oracle/ or the product."""
import math

M64 = (1 << 64) - 1


def wang64(key):
    key = (~key + (key << 21)) & M64
    key ^= key >> 24
    key = (key + (key << 3) + (key << 8)) & M64
    key ^= key >> 14
    key = (key + (key << 2) + (key << 4)) & M64
    key ^= key >> 28
    key = (key + (key << 31)) & M64
    return key


def idx_rho(h, p):
    idx = h >> (64 - p)
    w = (((h << 1) | 1) << (p - 1)) & M64
    rho = 64 - w.bit_length() + 1
    return idx, rho


def fold128(hi, lo):
    return lo ^ ((hi * 0x9E3779B97F4A7C15) & M64)


CODE = {ord("A"): 0, ord("a"): 0, ord("C"): 1, ord("c"): 1, ord("G"): 2, ord("g"): 2, ord("T"): 3, ord("t"): 3}


def records(fa: bytes):
    """The sequence strings of a FASTA / FASTQ buffer, read the way klib's kseq.h reads it (character by character, as
    kseq_read does with ks_getc / ks_getuntil): what is in front of the first '>' or '@' is skipped; name and comment run to
    the end of the line; sequence lines follow until a line starts with '>', '@' or '+'; a '+' line opens quality lines,
    read (one at least) until they are as long as the sequence, after which the reader again looks for '>' or '@' anywhere;
    a FASTQ record cut off inside its '+' line, or with a quality text of another length than its sequence, ends the reading
    (kseq_read's -2 under `while (kseq_read(ks) >= 0)`): that record and what follows are dropped."""
    out, i, n = [], 0, len(fa)
    last_char = 0
    while True:
        if last_char == 0:
            while i < n and fa[i] not in (62, 64):      # '>' '@'
                i += 1
            if i >= n:
                return out
            i += 1
        last_char = 0
        while i < n and fa[i] != 10:                    # header line
            i += 1
        i += 1
        seq = bytearray()
        c = -1
        while i < n:
            c = fa[i]
            i += 1
            if c in (62, 43, 64):                       # '>' '+' '@'
                break
            if c == 10:
                c = -1
                continue
            seq.append(c)
            while i < n and fa[i] != 10:                # the rest of the line
                seq.append(fa[i])
                i += 1
            i += 1
            if len(seq) > 1 and seq[-1] == 13:          # KS_SEP_LINE: one trailing '\r' goes
                seq.pop()
            c = -1
        if c in (62, 64):
            out.append(bytes(seq))
            last_char = c
            continue
        if c != 43:
            out.append(bytes(seq))
            return out
        while True:                                     # rest of the '+' line: ks_getc until '\n'
            if i >= n:
                return out                              # -1 inside the '+' line: kseq_read returns -2, the record is dropped
            ch = fa[i]
            i += 1
            if ch == 10:
                break
        qual = bytearray()
        while True:                                     # do { ks_getuntil2(line, append) } while (qual.l < seq.l)
            if i >= n:
                break                                   # nothing left to read: the call returns -1
            while i < n and fa[i] != 10:
                qual.append(fa[i])
                i += 1
            i += 1
            if len(qual) > 1 and qual[-1] == 13:
                qual.pop()
            if len(qual) >= len(seq):
                break
        if len(qual) != len(seq):
            return out                                  # -2: this record and everything behind it are not read
        out.append(bytes(seq))


def tokenize(fa: bytes):
    """-> list of tokens 0..3 / 4 (BREAK: an ambiguous sequence byte, and one at the end of every record)"""
    out = []
    for seq in records(fa):
        out.extend(CODE.get(c, 4) for c in seq)
        out.append(4)
    return out


def kmers(fa: bytes, k, canonical=True):
    """every (canonical) k-mer occurrence as a Python int"""
    toks = tokenize(fa)
    res = []
    run = 0
    for i, t in enumerate(toks):
        if t == 4:
            run = 0
            continue
        run += 1
        if run >= k:
            w = toks[i - k + 1: i + 1]
            f = 0
            for c in w:
                f = (f << 2) | c
            if canonical:
                r = 0
                for c in reversed(w):
                    r = (r << 2) | (3 - c)
                f = min(f, r)
            res.append(f)
    return res


def sketch(fa: bytes, k, p, canonical=True):
    regs = [0] * (1 << p)
    for x in kmers(fa, k, canonical):
        if k > 32:
            x = fold128(x >> 64, x & M64)
        idx, rho = idx_rho(wang64(x), p)
        if rho > regs[idx]:
            regs[idx] = rho
    return regs


def exact_count(fas, k, canonical=True):
    s = set()
    for fa in fas:
        s.update(kmers(fa, k, canonical))
    return len(s)


def ertl_mle(c, p):
    """Ertl 2017 Algorithm 8, transcribed from the paper's pseudo-code (floats are IEEE doubles)."""
    q = 64 - p
    m = 1 << p
    if c[q + 1] == m:
        return math.inf
    kmin = next(k for k in range(q + 2) if c[k])
    kminp = max(1, kmin)
    kmax = max(k for k in range(q + 2) if c[k])
    kmaxp = min(q, kmax)
    z = 0.0
    for k in range(kmaxp, kminp - 1, -1):
        z = 0.5 * z + c[k]
    z = math.ldexp(z, -kminp)
    cp = c[q + 1] + (c[kmaxp] if q >= 1 else 0)
    a = z + c[0]
    mp = m - c[0]
    b = z + math.ldexp(c[q + 1], -q)
    x = mp / (0.5 * b + a) if b <= 1.5 * a else (mp / b) * math.log1p(b / a)
    dx, gprev = x, 0.0
    relerr = 1e-2 / math.sqrt(m)
    while dx > x * relerr:
        kappam1 = math.frexp(x)[1]
        xp = math.ldexp(x, -max(kmaxp + 1, kappam1 + 2))
        xp2 = xp * xp
        h = xp - xp2 / 3.0 + (xp2 * xp2) * (1.0 / 45.0 - xp2 / 472.5)
        for _ in range(kappam1, kmaxp - 1, -1):
            hp = 1.0 - h
            h = (xp + h * hp) / (xp + hp)
            xp += xp
        g = cp * h
        for k in range(kmaxp - 1, kminp - 1, -1):
            hp = 1.0 - h
            h = (xp + h * hp) / (xp + hp)
            xp += xp
            g += c[k] * h
        g += x * a
        if gprev < g <= mp:
            dx *= (g - mp) / (gprev - g)
        else:
            dx = 0.0
        x += dx
        gprev = g
    return x * m
