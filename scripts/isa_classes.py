"""profiles/r04_isa_classes.json: the instruction classes of K1's hot loops, counted in the ISA of the shipped build.

    python scripts/isa_classes.py [--asm FILE] [--out profiles/r04_isa_classes.json]

Without --asm the script compiles dandd_amd/csrc/dd_sweep.hip with the flags of dandd_amd/build.py plus -save-temps
(hipcc cross-compiles gfx950 without a GPU) and reads the .s it leaves under build/isa/.

What it counts.  A kernel's loops are found from the labels and backward branches of its function body; one pass of a loop
is walked along its common path (hot_path: blocks guarded by `s_cbranch_execz` -- a register that must rise, the long form
of rho -- are skipped).  The HOT loop of a hashing kernel is the loop whose pass holds the hash of its window class (>= 2
v_mad_u64_u32 per hash: the multiplications by 2^21 - 1 and by 265 of Wang's mix) with the FEWEST vector instructions --
the BREAK-free variant, which ~96 % of waves run.  Its VALU instructions are split into the two issue classes measured on
gfx950 by scripts/ubench.hip (profiles/r01_ubench_issue_costs.txt, 4 waves per SIMD): the cheap class (v_xor / and / or /
not / mov / add_u32 / sub_u32 / lshrrev_b32) and everything else (shifts left, funnel shifts, multiplies, compares,
selects, ffbh, and every 64-bit operation).  bench.py prices the PMC-counted instructions per update
(profiles/r04_k1_counters_*.json) with this mix: valu_bound.frac_of_mix.
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHEAP = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32"}


def issue_costs(path):
    """ns per wave instruction and SIMD at 4 waves per SIMD, per class, from the ubench table (mean over the class)."""
    cheap, dear, ghz = [], [], []
    for line in open(path):
        m = re.match(r"clock probe: .* -> ([0-9.]+) GHz", line)
        if m:
            ghz.append(float(m.group(1)))
        m = re.match(r"(v_[a-z0-9_]+)(?: sh\d)?\s+([0-9.]+) ns\s+([0-9.]+) ns\s+([0-9.]+) ns", line)
        if not m or m.group(1) in ("v_cndmask_b32", "v_fma_f32"):   # (the select's figure is its dependence on vcc in that loop, not its issue cost)
            continue
        (cheap if m.group(1) in CHEAP else dear).append(float(m.group(4)))
    clock = sum(ghz) / len(ghz)
    return {"cheap_ns": sum(cheap) / len(cheap), "dear_ns": sum(dear) / len(dear), "clock_ghz_in_ubench": clock,
            "cheap_cycles": sum(cheap) / len(cheap) * clock, "dear_cycles": sum(dear) / len(dear) * clock,
            "cheap_ops": sorted(CHEAP), "from": "profiles/r01_ubench_issue_costs.txt (column 4 waves per SIMD)"}


def functions(asm):
    """mangled kernel name -> list of lines of its body"""
    out, name, body = {}, None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", line)
        if m:
            name, body = m.group(1), []
            out[name] = body
            continue
        if name is not None:
            body.append(line)
            if line.strip().startswith("s_endpgm"):
                name = None
    return out


def all_loops(body):
    """[(first line, last line)] of every loop: a label and the LAST later branch back to it"""
    labels = {}
    for i, line in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", line)
        if m:
            labels[m.group(1)] = i
    last = {}
    for i, line in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB[0-9_]+)", line)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            last[labels[m.group(1)]] = i
    return sorted(last.items())


def innermost_loops(body):
    """[(first line, last line)] of loops that contain no other loop: a label and a later branch back to it"""
    labels = {}
    for i, line in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", line)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, line in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB[0-9_]+)", line)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))
    return [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]


def hot_path(body, lo, hi, must=(), skip=None):
    """The lines one pass of the loop [lo, hi] executes when nothing rare happens: `s_cbranch_execz` (no lane needs the
    guarded block -- a register that must rise, the long form of rho) is TAKEN, every other conditional branch falls
    through unless it is the loop's back edge, unconditional branches are followed.  A guarded block that holds one of the
    memory operations in `must` is the update itself (e.g. the store behind `slot < capacity`) and is entered.  None if the
    walk leaves the loop."""
    labels = {}
    for i, line in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", line)
        if m:
            labels[m.group(1)] = i
    out, i, steps = [], lo, 0
    while steps < 20000:
        steps += 1
        if i < lo or i > hi:
            return None
        line = body[i]
        m = re.match(r"\s+(s_c?branch\S*)\s+(\.LBB[0-9_]+)", line)
        if not m:
            out.append(line)
            i += 1
            continue
        op, tgt = m.group(1), labels.get(m.group(2), -1)
        out.append(line)
        if op == "s_cbranch_execz":
            # the block this branch guards: what runs when it falls through, unconditional branches followed, up to the
            # next conditional branch
            guarded, j = [], i + 1
            while len(guarded) < 40 and 0 <= j < len(body):
                mb = re.match(r"\s+(s_c?branch\S*)\s+(\.LBB[0-9_]+)", body[j])
                if mb and mb.group(1) == "s_branch":
                    j = labels.get(mb.group(2), -1)
                    continue
                if mb:
                    break
                guarded.append(body[j])
                j += 1
        if op == "s_cbranch_execnz" and skip is not None:
            # (the compiler put the update BEHIND a taken branch: "some lane has a record" -> the block with the slot's atomic and
            # the store; follow it when what stands there, read through uniform vcc / scc branches, holds one of `must`)
            ahead, j = [], tgt
            while len(ahead) < 60 and 0 <= j < len(body):
                mb = re.match(r"\s+(s_c?branch\S*)\s+(\.LBB[0-9_]+)", body[j])
                if mb and mb.group(1) == "s_branch":
                    j = labels.get(mb.group(2), -1)
                    continue
                if mb and mb.group(1) in ("s_cbranch_execz", "s_cbranch_execnz"):
                    break
                if not mb:
                    ahead.append(body[j])
                j += 1
            if any(re.match(r"\s+" + pre, l) for l in ahead for pre in must):
                i = tgt
            else:
                i += 1
        elif op == "s_cbranch_execz" and skip is not None:
            # (a kernel whose update sits behind nested guards: every guarded block is entered except those that hold one of `skip`)
            if any(re.match(r"\s+" + pre, l) for l in guarded for pre in skip):
                if tgt == lo:
                    return out       # (the skipped block was the pass's last: back at the loop's head)
                i = tgt
            else:
                i += 1
        elif op == "s_cbranch_execz" and any(re.match(r"\s+" + pre, l) for l in guarded for pre in must):
            i += 1
        elif tgt == lo:
            return out           # back to the loop's head: one pass done
        elif op == "s_branch" or op == "s_cbranch_execz":
            i = tgt
        else:
            i += 1
    return None


def count(body, lo, hi):
    return count_lines(body[lo:hi + 1])


def count_lines(lines):
    c = {"valu_cheap": 0, "valu_dear": 0, "salu": 0, "lds": 0, "vmem": 0, "by_op": {}}
    for line in lines:
        m = re.match(r"\s+([a-z_0-9]+)", line)
        if not m:
            continue
        op = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", m.group(1))
        if op.startswith("v_"):
            c["valu_cheap" if op in CHEAP else "valu_dear"] += 1
            c["by_op"][op] = c["by_op"].get(op, 0) + 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
            c["salu"] += 1
    c["valu"] = c["valu_cheap"] + c["valu_dear"]
    return c


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_isa_classes.json"))
    a = ap.parse_args()
    asm_path = a.asm
    if not asm_path:
        sys.path.insert(0, ROOT)
        from dandd_amd import build as b
        d = os.path.join(ROOT, "build", "isa")
        os.makedirs(d, exist_ok=True)
        flags = [f for f in b.FLAGS if f not in ("-shared",)]
        subprocess.check_call([b.hipcc()] + flags + ["-save-temps", "-c", "-o", os.path.join(d, "dd_sweep.o"), os.path.join(b.CSRC, "dd_sweep.hip")], cwd=d)
        asm_path = os.path.join(d, "dd_sweep-hip-amdgcn-amd-amdhsa-gfx950.s")
    asm = open(asm_path).read()
    fns = functions(asm)
    pretty = demangle(list(fns))
    costs = issue_costs(os.path.join(ROOT, "profiles", "r01_ubench_issue_costs.txt"))
    # kernels whose inner loop hashes: name pattern -> updates one pass of the hot loop makes
    # (the last two fields: memory operations a pass of the real loop must hold -- a walk that skipped the update itself,
    # e.g. the `valid` guard of a BREAK-aware variant, is not the hot path -- and operations that mark a guarded block as
    # part of the update: the record's store behind `slot < capacity`)
    want = [
        (r"sweep_kernel<(\d), true>", 2, "registers in LDS (log2m <= 16): the k-pair loop of sweep_token, two updates per pass", ("ds_read",), ()),
        (r"scatter_kernel<(\d), true>", 1, "filtered epochs (log2m >= 17): one update per pass of the token loop", ("ds_read",), ()),
        (r"scatter_first_bin_kernel<(\d), true>", 1, "first epoch (log2m >= 17): one update per pass of the token loop -- the record's path (the rho = 1 lanes' ds_or block is the one skipped)", ("ds_add_rtn", "global_store"), (), ("ds_or",)),
    ]
    kc_name = {"0": "k <= 16 (32-bit windows)", "1": "k 17-32 (64-bit)", "3": "k 33-48 (96-bit)", "2": "k 49-64 (128-bit)"}
    out = {"made_by": "scripts/isa_classes.py over `hipcc -save-temps` of dandd_amd/csrc/dd_sweep.hip (the flags of dandd_amd/build.py)",
           "issue_costs": costs, "kernels": {}}
    for mangled, body in fns.items():
        name = re.sub(r"\(anonymous namespace\)::|dd::|void ", "", pretty[mangled]).split("(")[0]
        for pat, per_pass, what, must, enter, *rest in want:
            skip = rest[0] if rest else None
            m = re.fullmatch(pat, name)
            if not m:
                continue
            need_mads = 2 * per_pass
            cands = []
            for lo, hi in all_loops(body):
                path = hot_path(body, lo, hi, enter if skip is None else must, skip)
                if path is None:
                    continue
                c = count_lines(path)
                if not all(any(re.match(r"\s+" + pre, l) for l in path) for pre in must):
                    continue
                if c["by_op"].get("v_mad_u64_u32", 0) >= need_mads:
                    cands.append((c["valu"], lo, hi, c))
            if not cands:
                continue
            cands.sort(key=lambda t: t[0])
            _, lo, hi, c = cands[0]
            cyc = (c["valu_cheap"] * costs["cheap_cycles"] + c["valu_dear"] * costs["dear_cycles"]) / per_pass
            out["kernels"][name] = {
                "class": kc_name.get(m.group(1), m.group(1)), "what": what, "updates_per_pass": per_pass,
                "hot_loop_lines": [lo + 1, hi + 1], "loops_with_the_hash": len(cands),
                "valu_per_update": c["valu"] / per_pass, "valu_cheap_per_update": c["valu_cheap"] / per_pass,
                "valu_dear_per_update": c["valu_dear"] / per_pass, "salu_per_update": c["salu"] / per_pass,
                "lds_per_update": c["lds"] / per_pass, "vmem_per_update": c["vmem"] / per_pass,
                "cheap_fraction": c["valu_cheap"] / max(1, c["valu"]),
                "issue_cycles_per_update": cyc,
                "mean_cycles_per_valu": cyc / (c["valu"] / per_pass),
                "by_op": dict(sorted(c["by_op"].items(), key=lambda kv: -kv[1])),
            }
    json.dump(out, open(a.out, "w"), indent=1)
    for k, v in sorted(out["kernels"].items()):
        print(f'{k:60s} VALU/update {v["valu_per_update"]:5.1f} (cheap {v["valu_cheap_per_update"]:4.1f}, dear {v["valu_dear_per_update"]:4.1f})  '
              f'{v["issue_cycles_per_update"]:6.1f} cycles  LDS {v["lds_per_update"]:.1f} SALU {v["salu_per_update"]:.1f}')


if __name__ == "__main__":
    main()
