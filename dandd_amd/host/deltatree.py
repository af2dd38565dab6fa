"""DandD's experiment layer (delta trees, progressive unions, K-independent Jaccard) on top of the
MI355X sketching engine.

Behavioural mirror of /root/reference/lib/huffman_dandd.py and the SketchObj lifecycle of
/root/reference/lib/sketch_classes.py:124-373 -- same class and attribute names (the CLI pickles
these objects), same argmax-k hill-climb (ties move on, `kstart` is mutated as the climb proceeds,
lib/huffman_dandd.py:106-146), same tree shapes, same CSV rows -- with every `subprocess` call
replaced by a backend object (dandd_amd.host.backend.HipBackend: the GPU).  The goldens in
tests/golden/ref_*.json were produced by running the reference itself; tests/test_host_golden.py
replays them against this module.

Differences that are deliberate:
  * a leaf k-batch is ONE fused GPU pass over the FASTA instead of `parallel` over k;
  * `Sketch.cmd` holds a descriptive string instead of a shell line;
  * k < 1 is never sketched (the reference would create a k=-1 placeholder when a climb reaches k=0).
"""
import csv
import os
import weakref
import pickle
import sys
from itertools import permutations
from math import factorial
from random import sample, shuffle

from .store import Catalog, SketchPath, ensure_dir, forget_sketch, sketch_exists

# ---------------------------------------------------------------------------------------------
# backend plumbing: objects of this module are pickled, the GPU context is not
# ---------------------------------------------------------------------------------------------
_backend_factory = None
_backends = {}
RESIDENT = False    # `dandd serve` sets it: backends outlive commands, so what a command leaves on the device is worth keeping
_SWEPT = weakref.WeakKeyDictionary()   # DeltaTreeNode -> the ks its subtree has been brought up to date for, in this command


def new_command():
    """Forget what an earlier command of this process learned about the sketch directory (dandd serve calls it per command)."""
    _SWEPT.clear()
    from . import store
    store.new_command()
    for be in _backends.values():          # (registers of sketch files kept in memory: trusted only while size and mtime stand, but a
        forget = getattr(be, "new_command", None)   # command boundary is a good place to stop trusting altogether)
        if forget:
            forget()
_leaf_batch = None  # set while a tree solves its leaves: sketches missing ks for all of them at once


def set_backend_factory(factory):
    """factory(registers:int, canonicalize:bool) -> backend.  None restores the GPU default."""
    global _backend_factory
    _backend_factory = factory
    _backends.clear()


def _backend_key(experiment):
    exact = experiment.get("tool") == "kmc"
    return (int(experiment["registers"]), bool(experiment["canonicalize"]), exact)


def _make_backend(key):
    # both fail loudly without libdandd_hip.so / a gfx950 GPU; under torch.distributed.run every rank drives its own GPU
    from .backend import HipBackend, HipExactBackend
    cls = HipExactBackend if key[2] else HipBackend
    return cls(log2m=key[0], canonical=key[1], device=int(os.environ.get("LOCAL_RANK", "0")))


_prewarm = {}   # key -> (thread, result box)


def prewarm_backend(experiment, first_launches=True):
    """Start creating the GPU backend of `experiment` on a thread of its own: loading libdandd_hip.so, HIP's first
    use of the device and the first launch of every kernel module cost 0.25-0.35 s in a fresh process, during which a
    one-shot `dandd` command has blake2b digests, directory walks and imports of its own to do (ctypes calls release
    the GIL).  backend_for() picks the result up -- or the exception, which it re-raises.  (Measured on `tree` over
    10 x 50 Mbp, four alternating runs: 0.68 s with / 0.66 s without at log2m 14, 0.82 / 0.85 at log2m 20 -- within the
    box's noise; in `progressive` and `kij`, whose host-side set-up is short, it cost 0.05-0.1 s and is not used.)"""
    import threading
    key = _backend_key(experiment)
    if _backend_factory is not None or key in _backends or key in _prewarm or os.environ.get("DANDD_NO_PREWARM") == "1":
        return
    box = []

    def work():
        try:
            be = _make_backend(key)
            if first_launches and hasattr(be.engine, "warmup"):   # (the sketching kernels: a command that will sketch)
                be.engine.warmup()
            box.append(be)
        except BaseException as e:  # handed to the thread that asks for the backend
            box.append(e)

    t = threading.Thread(target=work, name="dandd-backend-prewarm")
    _prewarm[key] = (t, box)
    t.start()


def backend_for(experiment):
    key = _backend_key(experiment)
    if key in _prewarm:
        t, box = _prewarm.pop(key)
        t.join()
        if isinstance(box[0], BaseException):
            raise box[0]
        _backends[key] = box[0]
    if key not in _backends:
        _backends[key] = _backend_factory(key[0], key[1]) if _backend_factory is not None else _make_backend(key)
    if RESIDENT and hasattr(_backends[key], "resident"):
        _backends[key].resident = True
    return _backends[key]


# ---- one process per GPU (python -m torch.distributed.run ... -m dandd_amd.host.cli <subcommand> ...) ---
# Leaf sketching -- the only heavy step -- is sharded over the ranks by file size; the sketches travel
# the way they always do in DandD, as files in the shared sketch directory; after a barrier rank 0
# carries on alone with every leaf sketch cached and the other ranks are done.  The sharding is switched
# on by the CLI (set_dist_active) for the commands in which EVERY rank takes part, never by the mere
# presence of WORLD_SIZE in the environment: a barrier needs all its parties.
class WorkerDone(Exception):
    """Raised on ranks > 0 once their share of the leaf sketches is on disk."""


_dist_active = False


def set_dist_active(on):
    global _dist_active
    _dist_active = bool(on)


def dist_ranks():
    if not _dist_active:
        return 0, 1
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def dist_barrier():
    """The ONE meeting point of a multi-rank command: every rank's share of the leaf sketches is on disk.  Behind
    it rank 0 finishes alone, so the sharding is switched off here: from now on `_batch_leaf_sketch` batches ALL
    leaves again and the hill-climb's batch hook is back (a rank 0 that kept seeing world > 1 would batch only its
    own shard and sketch the other leaves one genome at a time)."""
    if not _dist_active:
        return
    try:
        import torch.distributed as dist
    except ImportError:
        return
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    set_dist_active(False)


class RowTable:
    """Rows of ONE shape held as columns: what a list of dicts holds, for tables of tens of thousands of rows that are made
    column-wise anyway (`kij --jaccard` over 64 genomes: 62 496 rows of 9 cells).  Iterating or indexing gives the dicts;
    write_listdict_to_csv writes the columns without ever making them."""

    def __init__(self, names):
        self.cols = {n: [] for n in names}

    def __len__(self):
        return len(next(iter(self.cols.values()))) if self.cols else 0

    def __iter__(self):
        names = list(self.cols)
        for values in zip(*self.cols.values()):
            yield dict(zip(names, values))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return list(self)[i]
        return {n: c[i] for n, c in self.cols.items()}


def write_listdict_to_csv(outfile, listdict, suffix="", last_col=None):
    """Rows as CSV, union of keys as header, `fastas`/`files` last (they may contain commas)."""
    names = sorted(listdict.cols) if isinstance(listdict, RowTable) else sorted({k for row in listdict for k in row})
    for special in ("fastas", "files"):
        if special in names:
            last_col = special
    if last_col in names:
        names.remove(last_col)
        names.append(last_col)
    out = sys.stdout if outfile in (None, "-") else open(outfile + suffix, "w", newline="")
    try:
        if isinstance(listdict, RowTable):
            # The csv module's "excel" dialect written by hand, a COLUMN at a time: a cell is str(value); it is quoted (quotes
            # doubled) iff it holds a comma, a quote or a line break; rows end in \r\n.  A column of few distinct values -- the
            # two paths and titles of a pair stand in each of its rows, a leaf's cardinality at k in every pair's row -- is
            # formatted once per value.  62 496 rows of 9 cells: 0.12 s through csv.writer, 0.03 s this way; same bytes
            # (tests/test_store.py::test_rowtable_csv_is_the_csv_modules).
            import re
            needs_quote = re.compile('[,"\r\n]').search

            def cell(v):
                t = v if isinstance(v, str) else ("" if v is None else str(v))
                return '"' + t.replace('"', '""') + '"' if needs_quote(t) else t
            text_cols = []
            for n in names:
                col = listdict.cols[n]
                kinds = set(map(type, col))
                # (by OBJECT, not by value: 0.0 == -0.0 == 0 and nan != nan; the repeats are the same objects)
                repeats = len(col) > 256 and 4 * len(set(map(id, col[:256]))) < 256    # (judged on the first 256 cells)
                if repeats:
                    memo = {i: cell(v) for i, v in {id(v): v for v in col}.items()}
                    text_cols.append(list(map(memo.__getitem__, map(id, col))))
                elif kinds <= {float, int}:
                    text_cols.append(list(map(str, col)))       # (numbers never need quoting)
                else:
                    text_cols.append([cell(v) for v in col])
            out.write(",".join(cell(n) for n in names) + "\r\n")
            if text_cols and text_cols[0]:
                out.write("\r\n".join(map(",".join, zip(*text_cols))) + "\r\n")
        elif len(names) > 1 and all(len(row) == len(names) for row in listdict):
            # every row has every column: the same bytes as DictWriter's, from C all the way (a `kij --jaccard` over 64
            # genomes writes 62 496 rows of 10 cells)
            from operator import itemgetter
            w = csv.writer(out)
            w.writerow(names)
            w.writerows(map(itemgetter(*names), listdict))
        else:
            w = csv.DictWriter(out, fieldnames=names)
            w.writeheader()
            w.writerows(listdict)
    finally:
        if out is not sys.stdout:
            out.close()


def permute(length, norder, preexist=frozenset(), exhaust=False, verbose=False):
    """`norder` distinct orderings of range(length), extending `preexist`
    (lib/huffman_dandd.py:36-60: shuffled full enumeration below 7!+1, rejection sampling above)."""
    total = factorial(length)
    result = set(preexist)
    norder = min(norder, total)
    if total < 5041 or norder == total or exhaust:
        pool = list(permutations(range(length)))
        shuffle(pool)
    else:
        pool = []
        while len(pool) < norder:
            pool.extend(tuple(sample(range(length), length)) for _ in range(norder))
            pool = list(set(pool))
    for cand in pool:
        if len(result) >= norder:
            break
        result.add(cand)
    return result


# ---------------------------------------------------------------------------------------------
class Sketch:
    """One (set of FASTAs, k) sketch: file on disk + cached cardinality (the reference's SketchObj)."""

    def __init__(self, kval, sfp, speciesinfo, experiment, presketches=()):
        self.kval = kval
        self.sfp = sfp
        self.sketch = None
        self.cmd = None
        self.card = 0
        self.delta_pos = 0
        self.speciesinfo = speciesinfo
        self.experiment = experiment
        self._presketches = list(presketches)
        experiment["baseset"].add(sfp.base)
        if kval > 0:
            self.create_sketch()
            self.card = self.check_cardinality()
            self.delta_pos = self.card / kval

    def __lt__(self, other):
        return self.delta_pos < other.delta_pos

    def __gt__(self, other):
        return self.delta_pos > other.delta_pos

    def __repr__(self):
        return f"['sketch loc: {self.sketch}', k: {self.kval}, pos delta: {self.delta_pos}, cardinality: {self.card}, command: {self.cmd}  ]"

    def sketch_check(self, path=None):
        return sketch_exists(path or self.sfp.full)

    def _build(self):
        be = backend_for(self.experiment)
        if self.sfp.ngen == 1:
            be.leaf(self.sfp.ffiles[0], [self.kval], [self.sfp.full])
        else:
            be.union(self._presketches, self.sfp.full)

    def create_sketch(self, just_do_it=False):
        if self.sfp.ngen < 1:
            raise RuntimeError("For some reason you are trying to sketch an empty list of files. Don't do that.")
        be = backend_for(self.experiment)
        op = "sketch" if self.sfp.ngen == 1 else "union"
        self.cmd = be.describe(op, k=self.kval, out=os.path.basename(self.sfp.full)) if hasattr(be, "describe") else op
        # a union whose cardinality was already computed by a batched GPU schedule
        # (prefetch_union_cards) is not materialised as a file, exactly like --lowmem; the
        # cardinality cache is asked before the file system
        trusted = self.experiment["lowmem"] or (self.sfp.ngen > 1 and self.experiment.get("prefetched"))
        if just_do_it:
            self._build()
        elif trusted and self.check_cardinality() > 0:
            pass
        elif not self.sketch_check():
            self._build()
        self.sketch = self.sfp.full
        return self.sketch

    def individual_card(self):
        if self.kval == 0:
            return
        be = backend_for(self.experiment)
        try:
            value = be.card(self.sfp.full)
        except Exception:
            print(f"Recreating sketch {self.sfp.full}")
            self.create_sketch(just_do_it=True)
            value = be.card(self.sfp.full)
        self.speciesinfo.cardkey[self.sfp.full] = float(value)

    def check_cardinality(self):
        key, cards = self.sfp.full, self.speciesinfo.cardkey
        if key not in cards and self.experiment["lowmem"]:
            return 0
        if key not in cards or cards[key] is None or float(cards[key]) == 0:
            if not self.sketch_check():
                return 0
            self.individual_card()
        self.card = float(cards[key])
        self.delta_pos = self.card / int(self.kval)
        return float(self.card)

    def remove_sketch(self):
        import glob
        pattern = self.sfp.full.replace("{}", "*") if self.kval == 0 else self.sfp.full
        for f in glob.glob(pattern):
            try:
                os.remove(f)
            except FileNotFoundError:
                pass
            forget_sketch(f)


class DashSketchObj(Sketch):
    """HyperLogLog sketch (the reference's class of this name, lib/sketch_classes.py:302)."""


class KMCSketchObj(Sketch):
    """Exact k-mer database (`--exact`; the reference's class of this name, lib/sketch_classes.py:377)."""


def sketch_class(experiment):
    return KMCSketchObj if experiment.get("tool") == "kmc" else DashSketchObj


# ---------------------------------------------------------------------------------------------
class DeltaTreeNode:
    def __init__(self, node_title, children, speciesinfo, experiment, progeny=None):
        self.experiment = experiment
        self.speciesinfo = speciesinfo
        self.children = children
        self.mink = self.maxk = 0
        if experiment["ksweep"] is not None:
            self.mink, self.maxk = experiment["ksweep"]
        self.bestk = 0
        self.delta = 0
        self.ksketches = [None] * max(100, self.maxk + 2)
        if progeny:
            self.node_title = node_title
            self.progeny = list(progeny)
        else:  # a leaf is its own progeny; its title is the file name without the last extension
            self.progeny = [self]
            self.fastas = [node_title]
            self.node_title = os.path.splitext(os.path.basename(node_title))[0]
        self.fastas = [leaf.fastas[0] for leaf in self.progeny]
        self.ngen = len(self.progeny)

    def __repr__(self):
        return f"{self.__class__.__name__}['{self.node_title}', k: {self.bestk}, delta: {self.delta}, ngen: {self.ngen}, children: {self.children!r} ]"

    def __lt__(self, other):
        return self.ngen < other.ngen

    def _grow(self, k):
        if k >= len(self.ksketches):
            self.ksketches.extend([None] * (k - len(self.ksketches) + 2))

    def _template(self):
        """k-placeholder SketchPath of this node's file set (the set never changes)."""
        t = self.__dict__.get("_tmpl")
        if t is None:
            t = self._tmpl = SketchPath(self.fastas, 0, self.speciesinfo, self.experiment)
        return t

    # ---- sketch files for a k range (the reference's `parallel` batch, lib/huffman_dandd.py:148-239)
    def ksweep_update_node(self, mink, maxk):
        mink, maxk = int(mink), int(maxk)
        self._grow(maxk)
        template = self._template()
        if self.ksketches[0] is None:
            self.ksketches[0] = sketch_class(self.experiment)(0, template, self.speciesinfo, self.experiment)
        # ks this node (and therefore its whole subtree) was already brought up to date for, in this
        # process: leaves are shared by hundreds of spiders and would otherwise be re-checked every time
        # (kept BESIDE the node, not in it: a memo inside the object went into the tree pickle -- with this process's pid in
        # it, so no two runs wrote the same bytes -- and a resident `dandd serve`, whose pid never changes, would have believed a
        # pickle's memo about sketch files of another day; new_command() forgets everything between commands)
        swept = _SWEPT.get(self)
        if swept is None:
            swept = _SWEPT[self] = set()
        ks = [k for k in range(max(1, mink), maxk + 1) if k not in swept]
        if not ks:
            return
        for k in ks:
            ensure_dir(template.dir.replace("{}", str(k)))
            self.experiment["baseset"].add(template.base.replace("{}", str(k)))
        for child in (self.children if self.ngen > 1 else []):
            child.ksweep_update_node(mink, maxk)
        cards = self.speciesinfo.cardkey
        todo = []
        for k in ks:
            path = template.with_k(k)
            # (dictionary first: a prefetched union has a cardinality and no file, and asking the
            # file system about every such path dominated `progressive`)
            if (self.experiment["lowmem"] or self.experiment.get("prefetched")) and self.ngen > 1 \
                    and float(cards.get(path) or 0) > 0:
                continue
            if sketch_exists(path):
                continue
            todo.append(k)
        if todo and self.ngen == 1 and _leaf_batch is not None:
            _leaf_batch(todo)  # the same ks (and a margin) for every leaf of the tree in one batch
            todo = [k for k in todo if not sketch_exists(template.with_k(k))]
        if todo:
            be = backend_for(self.experiment)
            if self.ngen == 1:
                be.leaf(self.fastas[0], todo, [template.with_k(k) for k in todo])  # one fused GPU pass
            else:
                for k in todo:
                    ins = [c.ksketches[0].sfp.with_k(k) for c in self.children]
                    be.union(ins, template.with_k(k))
        swept.update(ks)

    def update_node(self, kval):
        """Sketch object (file + cardinality) for k at this node and, first, at every descendant."""
        window = self.experiment["ksweep"]
        if window is not None and not (window[0] <= kval <= window[1]):
            print(f"k={kval} is outside of ksweep range ", window)
            return
        self._grow(kval)
        if self.ksketches[kval]:
            return
        sfp = self._template().at_k(kval, self.speciesinfo, self.experiment)
        inputs = []
        if self.ngen > 1:
            for child in self.children:
                child.update_node(kval)
                inputs.append(child.ksketches[kval].sketch)
        self.ksketches[kval] = sketch_class(self.experiment)(kval, sfp, self.speciesinfo, self.experiment, presketches=inputs)

    def node_ksweep(self, mink, maxk):
        self.ksweep_update_node(mink, maxk)
        for k in range(max(1, mink), maxk + 1):
            if self.ksketches[k] is None:
                self.update_node(k)
        self.mink, self.maxk = mink, maxk

    # ---- argmax-k local search (lib/huffman_dandd.py:106-146) --------------------------------------
    def find_delta_helper(self, kval, direction=1):
        if self.experiment["tool"] == "dashing" and kval > 32 and not self.experiment.get("allow_k64"):
            raise ValueError("Exploratory k value is too high for dashing. Either something is amiss "
                             "with your data or you need to be using --exact mode")
        if kval < 1:
            return
        self._grow(kval + 1)
        self.node_ksweep(mink=kval - 1, maxk=kval + 1)
        self.update_node(kval)
        if direction < 0:
            self.mink = kval
        else:
            self.maxk = kval
        candidate = self.ksketches[kval].delta_pos
        if self.delta <= candidate:  # ties keep climbing
            self.speciesinfo.kstart = kval
            self.bestk = kval
            self.delta = candidate
            self.find_delta_helper(kval + direction, direction)

    def find_delta(self, kval):
        self.find_delta_helper(kval, 1)
        self.find_delta_helper(kval, -1)
        self.card = self.ksketches[self.bestk].card

    def summarize(self, mink=0, maxk=0, ordering_number=0):
        rows = []
        for k in range(mink, maxk + 1):
            s = self.ksketches[k]
            rows.append({"ngen": self.ngen, "kval": k, "card": s.card, "delta_pos": s.delta_pos,
                         "title": self.node_title, "command": s.cmd, "ordering": ordering_number})
        return rows


# ---------------------------------------------------------------------------------------------
class _FlatUnion:
    """The body node of a SubSpider as NUMBERS.  When a whole union schedule has come back from the GPU as one table of
    cardinalities (prefetch_union_cards: every pair x k of `kij`, every prefix x k of `progressive`), a SubSpider per pair or
    prefix -- a DeltaTreeNode, a Sketch object and a SketchPath per (set, k), 76 608 of them for 64 genomes -- computes nothing:
    it walks dictionaries.  This class goes through the same steps on the table itself -- find_delta / find_delta_helper with
    their `<=` ties and the kstart they move (lib/huffman_dandd.py:106-146), node_ksweep's k-1..k+1 window (:117), fill_tree's
    update_node of every bestk (:451-457), summarize (:289-301) -- and leaves the same traces a tree save reads: the base names in
    the experiment's base set, their sketchinfo entries, the ngen*/k* directories.  A k outside the table goes through the
    file-based backend (leaf sketches, union, card) as the object path does."""

    def __init__(self, tree, kids, experiment, table_row, lo, hi):
        sp = tree.speciesinfo
        self.sp, self.experiment, self.kids = sp, experiment, list(kids)
        self.row, self.lo, self.hi = table_row, int(lo), int(hi)
        self.fastas = [leaf.fastas[0] for c in self.kids for leaf in c.progeny]
        self.ngen = len(self.fastas)
        self.title = "_".join(os.path.basename(c.node_title) for c in self.kids)
        self.tmpl = SketchPath(self.fastas, 0, sp, experiment)   # (the '{}' form: registers its base as the node's placeholder sketch does)
        experiment["baseset"].add(self.tmpl.base)
        self.delta, self.bestk, self.mink, self.maxk = 0, 0, 0, 0
        self._touched = set()
        self._be = backend_for(experiment)
        self._base = self.tmpl.base.split("{}")       # the base name around its k

    def touch(self, k):
        """what update_node(k) leaves behind besides the Sketch object: the base name in the experiment's base set and its
        sketchinfo entry (a tree save lists them: <prefix>_sketchdb.txt).  NOT the ngen<N>/k<K> directory the reference makes
        for every k it looks at (lib/huffman_dandd.py:171-174): no file of this union is written unless card() has to build
        one, which makes the directory then -- 2 046 empty directories were 0.1 s of a 64-genome `progressive`."""
        if k in self._touched:
            return
        self._touched.add(k)
        base = str(k).join(self._base)
        self.experiment["baseset"].add(base)
        if base not in self.sp.sketchinfo:
            self.sp.sketchinfo[base] = {"sketchbase": base, "files": self.tmpl.files, "ngen": self.ngen, "kval": k,
                                        "registers": self.experiment["registers"]}

    def card(self, k):
        if self.row is not None and self.lo <= k <= self.hi:
            return float(self.row[k - self.lo])
        path = self.tmpl.with_k(k)
        cards = self.sp.cardkey
        if float(cards.get(path) or 0) > 0:
            return float(cards[path])
        # outside the table: the files, as DeltaTreeNode.ksweep_update_node + Sketch would (a hill-climb that leaves the window)
        for c in self.kids:
            c.ksweep_update_node(k, k)
        if self.ngen > 1 and not sketch_exists(path):
            ensure_dir(self.tmpl.dir.replace("{}", str(k)))
            self._be.union([c.ksketches[0].sfp.with_k(k) for c in self.kids], path)
        cards[path] = float(self._be.card(path))
        return cards[path]

    def command(self, k):
        op = "sketch" if self.ngen == 1 else "union"
        return self._be.describe(op, k=k, out=os.path.basename(self.tmpl.with_k(k))) if hasattr(self._be, "describe") else op

    def node_ksweep(self, mink, maxk):
        lo = max(1, mink)
        if maxk - lo >= 4:
            # a whole window at once (every pair of `kij --jaccard`, every prefix of a k-sweep `progressive`): the same traces as
            # touch(k) for each k, without a call per k
            pre, post = self._base
            seen = self._touched
            ks = [k for k in range(lo, maxk + 1) if k not in seen]
            bases = [pre + str(k) + post for k in ks]
            self._touched.update(ks)
            self.experiment["baseset"].update(bases)
            info, files, ngen, regs = self.sp.sketchinfo, self.tmpl.files, self.ngen, self.experiment["registers"]
            for k, base in zip(ks, bases):
                if base not in info:
                    info[base] = {"sketchbase": base, "files": files, "ngen": ngen, "kval": k, "registers": regs}
        else:
            for k in range(lo, maxk + 1):
                self.touch(k)
        self.mink, self.maxk = mink, maxk

    def find_delta_helper(self, kval, direction):
        while True:
            if self.experiment["tool"] == "dashing" and kval > 32 and not self.experiment.get("allow_k64"):
                raise ValueError("Exploratory k value is too high for dashing. Either something is amiss "
                                 "with your data or you need to be using --exact mode")
            if kval < 1:
                return
            self.node_ksweep(kval - 1, kval + 1)
            if direction < 0:
                self.mink = kval
            else:
                self.maxk = kval
            candidate = self.card(kval) / kval
            if not self.delta <= candidate:  # ties keep climbing
                return
            self.sp.kstart = kval
            self.bestk = kval
            self.delta = candidate
            kval += direction

    def find_delta(self, kval):
        self.find_delta_helper(kval, 1)
        self.find_delta_helper(kval, -1)

    def fill(self):
        """SubSpider.fill_tree in a hill-climb run: the root brought up to date at every node's argmax-k"""
        for k in sorted({c.bestk for c in self.kids} | {self.bestk}):
            if k > 0:
                self.touch(k)

    def summarize(self, mink, maxk, ordering_number):
        rows = []
        for k in range(mink, maxk + 1):
            if k == 0:   # the placeholder sketch (a hill-climb run's summary is made of these: SURVEY.md section 9)
                rows.append({"ngen": self.ngen, "kval": 0, "card": 0, "delta_pos": 0, "title": self.title, "command": None,
                             "ordering": ordering_number})
                continue
            c = self.card(k)
            rows.append({"ngen": self.ngen, "kval": k, "card": c, "delta_pos": c / k, "title": self.title,
                         "command": self.command(k), "ordering": ordering_number})
        return rows


# ---------------------------------------------------------------------------------------------
DEFAULT_EXPERIMENT = {"tool": "dashing", "registers": 20, "canonicalize": True, "debug": False, "nthreads": 0,
                      "baseset": set(), "safety": False, "fast": False, "verbose": False, "ksweep": None,
                      "lowmem": False}


class DeltaTree:
    def __init__(self, fasta_files, speciesinfo, nchildren=2, leafnodes=(), experiment=None, padding=True):
        self.experiment = experiment if experiment is not None else dict(DEFAULT_EXPERIMENT, baseset=set())
        self.mink = self.maxk = 0
        if self.experiment["ksweep"] is not None:
            self.mink, self.maxk = self.experiment["ksweep"]
        self.kstart = speciesinfo.kstart
        self.speciesinfo = speciesinfo
        if self.experiment["verbose"]:
            print("Now making tree for fastas: " + ", ".join(fasta_files))
        self._build_tree(fasta_files, nchildren)
        self.fill_tree(padding=padding)
        self.ngen = len(fasta_files)
        self.root = self._dt[-1]
        self.delta = self.root_delta()
        self.fastas = fasta_files
        if self.experiment["ksweep"] is None:
            speciesinfo.kstart = self.root_k()
        speciesinfo.save_references(fast=self.experiment["fast"])
        speciesinfo.save_cardkey(tool=self.experiment["tool"])

    def __sub__(self, other):
        print("Larger Tree Delta: ", self.delta)
        print("Subtree Delta: ", other.delta)
        print("Subtraction Result: ", self.delta - other.delta)
        return self.delta - other.delta

    def __repr__(self):
        return f"{self.__class__.__name__}(FASTAS: {self.fastas}, NODES: {self._dt[-1]!r})"

    def root_delta(self):
        return self._dt[-1].delta

    def root_k(self):
        return self._dt[-1].bestk

    def _solve(self, node):
        if self.experiment["ksweep"] is None:
            node.find_delta(self.speciesinfo.kstart)
        else:
            node.node_ksweep(mink=self.mink, maxk=self.maxk)

    def _predigest(self, leaves):
        """blake2b of every leaf file not yet in the catalog, hashed on a few threads (hashlib releases
        the GIL) instead of one file at a time inside SketchPath: 0.5 s -> 0.1 s for 10 x 50 MB."""
        from concurrent.futures import ThreadPoolExecutor
        from .store import file_digest
        missing = [leaf.fastas[0] for leaf in leaves
                   if os.path.basename(leaf.fastas[0]) not in self.speciesinfo.fastahex]
        if len(missing) < 2:
            return
        with ThreadPoolExecutor(max_workers=min(8, len(missing))) as pool:
            for path, digest in zip(missing, pool.map(file_digest, missing)):
                self.speciesinfo.fastahex[os.path.basename(path)] = digest

    def _batch_leaf_sketch(self, leaves, lo, hi):
        """Sketch files of k in [lo, hi] for every leaf that lacks any of them, by ONE pipelined batch
        (files are read/inflated ahead while the GPU sketches); sharded over the ranks when several."""
        be = backend_for(self.experiment)
        lo = max(int(lo), 1)
        # (a hill-climb never asks Dashing for k > 32, lib/huffman_dandd.py:109; an explicit --ksweep range is taken
        # as given, there as here -- the per-leaf path sketches those ks too, one genome at a time)
        if self.experiment["tool"] == "dashing" and not self.experiment.get("allow_k64") and self.experiment["ksweep"] is None:
            hi = min(int(hi), 32)
        if hi < lo or not hasattr(be, "leaf_many"):
            return 0
        rank, world = dist_ranks()
        if world > 1:
            # the plan covers ALL leaves and depends on nothing a rank could see differently (file sizes,
            # not which sketches happen to exist at the moment a rank looks): every rank computes the same
            # shards, then skips what is already on disk within its own
            from ..dist import shard_by_weight
            mine = set(shard_by_weight([os.path.getsize(leaf.fastas[0]) for leaf in leaves], world)[rank])
        todo, templates = [], []
        for i, leaf in enumerate(leaves):
            if world > 1 and i not in mine:
                continue
            tmpl = leaf._template() if hasattr(leaf, "_template") else SketchPath(leaf.fastas, 0, self.speciesinfo, self.experiment)
            if any(not sketch_exists(tmpl.with_k(k)) for k in range(lo, hi + 1)):
                for k in range(lo, hi + 1):
                    ensure_dir(tmpl.dir.replace("{}", str(k)))
                todo.append(leaf.fastas[0])
                templates.append(tmpl)
        if todo and (world > 1 or len(todo) > 1):
            cards = be.leaf_many(todo, lo, hi, lambda i, k: templates[i].with_k(k))
            # (the product backend estimates every sketch of the batch in one launch: its answers go where
            # individual_card would have put them one file at a time; an empty sketch's 0 is left for that path)
            if cards:
                self.speciesinfo.cardkey.update({p: c for p, c in cards.items() if c > 0})
        return len(todo)

    def presketch_range(self, lo, hi):
        """Sketch files of k in [lo, hi] for every leaf of this tree in one batch -- sharded over the ranks
        of a multi-GPU run, which meet at a barrier afterwards (every rank must call this)."""
        leaves = self.leaf_nodes()
        self._predigest(leaves)
        n = self._batch_leaf_sketch(leaves, lo, hi)
        dist_barrier()
        return n

    def _presketch_leaves(self, leaves, radius=3):
        """Leaf sketches for the ks the per-leaf searches are about to ask for -- the whole ksweep
        range, or kstart +- radius for the hill-climb -- made for ALL leaves by one pipelined batch
        instead of genome by genome (the reference's loop, lib/huffman_dandd.py:402-407, is strictly
        sequential).  Purely a cache warm-up: see _leaf_batch_hook for searches that leave the window."""
        be = backend_for(self.experiment)
        if not hasattr(be, "leaf_many") or len(leaves) < 2 or os.environ.get("DD_NO_PREFETCH"):
            return
        self._predigest(leaves)
        if self.experiment["ksweep"] is not None:
            lo, hi = (int(v) for v in self.experiment["ksweep"])
        else:
            lo, hi = max(1, int(self.speciesinfo.kstart) - radius), int(self.speciesinfo.kstart) + radius
        self._batch_leaf_sketch(leaves, lo, hi)

    def _leaf_batch_hook(self, leaves, margin=2):
        """While the leaves are being solved: a hill-climb that asks one leaf for ks outside what is on
        disk gets them -- and `margin` more on either side -- sketched for EVERY
        leaf in one batch, because the other leaves' searches are about to walk the same way.  One pass
        over all files per window extension instead of one file read per (leaf, step)."""
        if len(leaves) < 2 or os.environ.get("DD_NO_PREFETCH") or dist_ranks()[1] > 1:
            return None

        def hook(ks):
            return self._batch_leaf_sketch(leaves, min(ks) - margin, max(ks) + margin)
        return hook

    def _build_tree(self, symbol, nchildren, leafnodes=()):
        """Leaves first (in the order given; the sort by ngen is stable), then unions of `nchildren`
        consecutive nodes, each new union inserted behind the nodes that are not larger than it
        (lib/huffman_dandd.py:377-438, including its end-of-list widening of the last union)."""
        nodes = list(leafnodes) or [DeltaTreeNode(s, [], self.speciesinfo, self.experiment) for s in symbol]
        nodes.sort()
        self._presketch_leaves(nodes)
        if not leafnodes and dist_ranks()[1] > 1:  # a tree built from FASTA names in a multi-rank run
            rank = dist_ranks()[0]
            dist_barrier()
            if rank != 0:
                raise WorkerDone()
        global _leaf_batch
        _leaf_batch = self._leaf_batch_hook(nodes) if self.experiment["ksweep"] is None else None
        try:
            for leaf in nodes:
                self._solve(leaf)
        finally:
            _leaf_batch = None
        self._dt = nodes
        insert_at = 0
        current = 0
        while current != len(self._dt) - 1:
            step = nchildren - 1
            kids = self._dt[current:current + nchildren]
            union = DeltaTreeNode("_".join(k.node_title for k in kids), kids, self.speciesinfo, self.experiment,
                                  progeny=[leaf for k in kids for leaf in k.progeny])
            self._solve(union)
            while insert_at < len(self._dt) - step and self._dt[insert_at + step].ngen <= union.ngen:
                insert_at += step
            cut = insert_at + step
            self._dt = self._dt[:cut] + [union] + self._dt[cut:]
            current += nchildren
            if cut > len(self._dt) - 1:
                nchildren = len(self._dt) - current

    def print_list(self):
        print(" -> ".join(f"'{n.node_title}'({n.ngen}'({' '.join(p.node_title for p in n.progeny)})" for n in self._dt))

    def fill_tree(self, padding=False):
        root = self._dt[-1]
        if self.experiment["ksweep"] is None:
            for k in sorted({n.bestk for n in self._dt} - {0}):
                root.update_node(k)
        else:
            self.ksweep(*self.experiment["ksweep"])

    def leaf_nodes(self):
        return [n for n in self._dt if n.ngen == 1]

    def delete_sketches(self):
        for node in self._dt[:-1]:
            if node.ngen > 1:
                for s in node.ksketches:
                    if s is not None:
                        s.remove_sketch()

    def make_prefix(self, tag, label="", outdir=None):
        outdir = outdir or os.getcwd()
        label = "_" + label if label else ""
        return os.path.join(outdir, f"{tag}{label}_{self.ngen}_{self.experiment['tool']}")

    def save(self, fileprefix, fast=False):
        filepath = fileprefix + "_dtree.pickle"
        if not fast:
            from .compat import dump_tree
            with open(filepath, "wb") as f:
                dump_tree(self, f)  # under the reference's GLOBAL names: its own progressive/kij can load it
            print("Tree Pickle saved to: " + filepath)
            mapping = fileprefix + "_sketchdb.txt"
            write_listdict_to_csv(mapping, [self.speciesinfo.sketchinfo[b] for b in self.experiment["baseset"]])
            print(f"Output Sketch/DB mapping saved to {mapping}.")
        deltapath = fileprefix + "_deltas.csv"
        write_listdict_to_csv(deltapath, self.report_deltas())
        print("Deltas saved to: " + deltapath)
        return filepath

    def report_deltas(self):
        rows = []

        def visit(node):
            best = node.ksketches[node.bestk]
            rows.append({"delta": node.delta, "k": node.bestk, "title": node.node_title, "ngen": node.ngen,
                         "sketchloc": best.sketch, "card": best.card, "fastas": "|".join(node.fastas)})
            for child in node.children or []:
                visit(child)

        visit(self._dt[-1])
        return rows

    def nodes_from_fastas(self, fasta_list):
        return [n for n in self.leaf_nodes() if n.fastas[0] in fasta_list]

    def find_delta_delta(self, fasta_subset):
        rest = [f for f in self.fastas if f not in fasta_subset]
        small = SubSpider(self.nodes_from_fastas(rest), self.speciesinfo, self.experiment)
        print("Full Tree Delta: ", self.delta)
        print("Subtree Delta: ", small.delta)
        return self - small

    def ksweep(self, mink, maxk):
        for node in self._dt:
            node.node_ksweep(mink=mink, maxk=maxk)

    # ---- batched GPU union schedules ------------------------------------------------------------------
    def _leaf_files(self, leaves, lo, hi):
        """Make sure every leaf has its sketch file for k in [lo, hi] (one fused GPU pass per leaf)
        and return the paths as [leaf][k]."""
        rows = []
        for leaf in leaves:
            before = set(leaf.experiment["baseset"])
            leaf.ksweep_update_node(lo, hi)
            # ksweep_update_node notes every k of the range in the experiment's base set, as the reference's does
            # (lib/huffman_dandd.py:174) -- there every such k is then visited (node_ksweep) and registered.  This
            # window is a prefetch of OURS: a k the search never visits must not reach `save`, whose sketch/DB table
            # (lib/huffman_dandd.py:503) looks every base up and would die on it (hill-climb `progressive` did).
            # In a hill-climb run the reference never calls it at all: nothing of the window stays on the record.
            fresh = leaf.experiment["baseset"] - before
            if leaf.experiment.get("ksweep") is not None:
                fresh = {b for b in fresh if b not in self.speciesinfo.sketchinfo}
            leaf.experiment["baseset"].difference_update(fresh)
            tmpl = leaf.ksketches[0].sfp
            rows.append([tmpl.with_k(k) for k in range(lo, hi + 1)])
        return rows

    def prefetch_union_cards(self, groups, lo, hi, experiment, schedule=False):
        """Cardinalities of the unions `groups` (lists of leaf nodes) for k in [lo, hi], computed by
        ONE batched GPU launch per schedule instead of one union + one card per (set, k), and stored
        in the cardinality cache under the names the union sketches would have.  A backend without
        batch entry points (the CPU checkers used in tests) makes this a no-op.
        schedule=True: the table itself comes back -- {"table", "index" (id(leaf) -> row), "lo", "hi", "orders" (ordering as a
        tuple of rows -> its number in the table)} or None -- and the CALLER stores what it uses in the cache (the summaries
        below work on the table: _FlatUnion)."""
        be = backend_for(experiment)
        if lo < 1 or hi < lo or not groups or os.environ.get("DD_NO_PREFETCH"):
            return None if schedule else 0
        pair_mode = all(len(g) == 2 for g in groups)
        if not hasattr(be, "pairwise_cards" if pair_mode else "progressive_cards"):
            return None if schedule else 0
        leaves = []
        seen = set()
        for g in groups:
            for leaf in g:
                if id(leaf) not in seen:
                    seen.add(id(leaf))
                    leaves.append(leaf)
        index = {id(leaf): i for i, leaf in enumerate(leaves)}
        paths = self._leaf_files(leaves, lo, hi)
        cards = self.speciesinfo.cardkey
        filled = 0
        if pair_mode:
            table = be.pairwise_cards(paths)
            if schedule:
                experiment["prefetched"] = True
                return {"table": table, "index": index, "lo": lo, "hi": hi, "orders": {}}
            for a, b in groups:
                tmpl = SketchPath([a.fastas[0], b.fastas[0]], 0, self.speciesinfo, experiment)
                row = table[index[id(a)], index[id(b)]]
                for kk, k in enumerate(range(lo, hi + 1)):
                    cards[tmpl.with_k(k)] = float(row[kk])
                    filled += 1
        else:
            # every group is treated as an ordering; prefixes of length >= 2 are unions
            n = len(leaves)
            ords, used = [], []
            for g in groups:
                if len(g) == n:
                    ords.append([index[id(leaf)] for leaf in g])
                    used.append(g)
            if not ords:
                return None if schedule else 0
            table = be.progressive_cards(paths, ords)
            if schedule:
                experiment["prefetched"] = True
                return {"table": table, "index": index, "lo": lo, "hi": hi, "orders": {tuple(o): i for i, o in enumerate(ords)}}
            for o, g in enumerate(used):
                for j in range(1, n):
                    fastas = [leaf.fastas[0] for leaf in g[: j + 1]]
                    tmpl = SketchPath(fastas, 0, self.speciesinfo, experiment)
                    for kk, k in enumerate(range(lo, hi + 1)):
                        cards[tmpl.with_k(k)] = float(table[o, j, kk])
                        filled += 1
        experiment["prefetched"] = True
        return filled

    # ---- progressive unions (lib/huffman_dandd.py:574-663) -------------------------------------------
    def progressive_fastas(self, flist_loc=None):
        """The FASTAs a `progressive` run covers, in its order (lib/huffman_dandd.py:574-588)."""
        fastas = self.fastas
        fastas.sort()
        if flist_loc:
            with open(flist_loc) as f:
                wanted = [line.strip() for line in f]
            present = set(fastas)
            fastas = [f for f in wanted if f in present]
        return fastas

    def orderings_list(self, ordering_file=None, flist_loc=None, count=0):
        fastas = self.progressive_fastas(flist_loc)
        if count == 1:
            return fastas, [tuple(range(len(fastas)))]
        default = os.path.join(self.speciesinfo.sketchdir, f"{self.speciesinfo.tag}_{len(fastas)}_orderings.pickle")
        ordering_file = ordering_file or default
        orderings = set()
        if os.path.exists(ordering_file):
            with open(ordering_file, "rb") as f:
                orderings = pickle.load(f)
            if count == 0:
                return fastas, list(orderings)
            if count <= len(orderings):
                return fastas, list(orderings)[:count]
        elif count < 1:
            raise ValueError("You must provide a value for count when there is no default ordering file")
        orderings = permute(len(fastas), count, preexist=orderings, verbose=self.experiment["verbose"])
        with open(ordering_file, "wb") as f:
            pickle.dump(orderings, f)
        return fastas, list(orderings)

    def progressive_wrapper(self, flist_loc=None, count=30, ordering_file=None, step=1, debug=False):
        fastas, orderings = self.orderings_list(ordering_file=ordering_file, flist_loc=flist_loc, count=count)
        return self.progressive_union(flist=fastas, orderings=orderings, step=step)

    def progressive_union(self, flist, orderings, step):
        spider = DeltaSpider(fasta_files=flist, speciesinfo=self.speciesinfo, experiment=self.experiment)
        # all prefix unions of all orderings in one GPU launch (running max == flat union)
        sched = None
        if step == 1 and len(flist) > 1 and not self.experiment.get("safety"):
            by_fasta = {leaf.fastas[0]: leaf for leaf in spider.leaf_nodes()}
            if self.experiment["ksweep"] is not None:
                lo, hi = self.experiment["ksweep"]
            else:  # the hill-climbs stay within a few k of the leaves' and the root's argmax
                lo = max(1, min(leaf.bestk for leaf in by_fasta.values()) - 2)
                hi = spider.root.bestk + 3
                if self.experiment["tool"] == "dashing":
                    hi = min(hi, 32)
            groups = [[by_fasta[spider.fastas[j]] for j in ordering] for ordering in orderings]
            sched = spider.prefetch_union_cards(groups, int(lo), int(hi), self.experiment, schedule=True)
        results, summary = [], []
        for i, ordering in enumerate(orderings):
            if self.experiment["verbose"]:
                print(f"Now sweeping for ordering {i + 1}")
            rows, srows = spider.sketch_ordering(ordering, ordering_number=i + 1, step=step, schedule=sched)
            results.extend(rows)
            summary.extend(srows)
            self.speciesinfo.save_references(fast=self.experiment["fast"])
            self.speciesinfo.save_cardkey(tool=self.experiment["tool"], fast=self.experiment["fast"])
        self.experiment.pop("prefetched", None)  # the trust flag must not outlive this run (the tree is pickled)
        return results, summary

    def sketch_ordering(self, ordering, ordering_number, step=1, schedule=None):
        """Flat union of every prefix of the ordering (NOT previous union + one)."""
        krange = self.experiment["ksweep"] or (self.mink, self.maxk)
        lo, hi = int(krange[0]), int(krange[1])
        rows, summary = [], []
        leaves = self.leaf_nodes()
        # the schedule's table holds this ordering: every prefix is a row of numbers, not a SubSpider of objects (_FlatUnion)
        o = None
        if schedule is not None and step == 1:
            by_fasta = {leaf.fastas[0]: leaf for leaf in leaves}
            key = tuple(schedule["index"].get(id(by_fasta.get(self.fastas[j]))) for j in ordering)
            o = schedule["orders"].get(key)
        for i in range(1, len(ordering) + 1):
            if i % step:
                continue
            prefix = [self.fastas[j] for j in ordering[:i]]
            if o is not None:
                inside = set(prefix)
                kids = [leaf for leaf in leaves if leaf.fastas[0] in inside]      # tree order, as nodes_from_fastas gives them
                sub = _FlatUnion(self, kids, self.experiment, schedule["table"][o, i - 1], schedule["lo"], schedule["hi"])
                if i > 1 and sub.row is not None:   # what the union sketches' cardinalities would have been cached as
                    cards = self.speciesinfo.cardkey
                    for kk, k in enumerate(range(sub.lo, sub.hi + 1)):
                        cards[sub.tmpl.with_k(k)] = float(sub.row[kk])
                if i == 1:
                    sub.row = None                   # a single leaf: its own sketch files and cached cardinalities
                if self.experiment["ksweep"] is None:
                    sub.find_delta(self.speciesinfo.kstart)
                    sub.fill()
                    delta = sub.delta
                else:
                    sub.node_ksweep(lo, hi)
                    delta = None
                rows.append({"ngen": i, "kval": sub.bestk, "delta": delta, "ordering": ordering_number, "fastas": prefix})
                summary.extend(sub.summarize(lo, hi, ordering_number))
                continue
            sub = SubSpider(self.nodes_from_fastas(prefix), self.speciesinfo, self.experiment)
            sub.ksweep(mink=lo, maxk=hi)
            rows.append({"ngen": i, "kval": sub.root_k(), "delta": sub.delta, "ordering": ordering_number,
                         "fastas": prefix})
            summary.extend(sub.root.summarize(mink=lo, maxk=hi, ordering_number=ordering_number))
        return rows, summary

    # ---- K-independent Jaccard (lib/huffman_dandd.py:666-695) -----------------------------------------
    def pairwise_spiders(self, sublist=(), mink=0, maxk=0, jaccard=True):
        leaves = list(sublist) or self.leaf_nodes()
        pair_exp = dict(self.experiment)
        pair_exp.update({"fast": True, "safe": False, "ksweep": None})
        if jaccard and (mink == 0 or maxk == 0):
            if self.experiment["ksweep"]:
                mink, maxk = self.experiment["ksweep"]
                print("WARNING: If EITHER minimum OR maximum k are not provided with --mink and --maxk flags, "
                      "DandD will default to the --ksweep values embedded in the delta-tree input.")
            else:
                print("WARNING: If BOTH minimum AND maximum k are not provided either by the input delta-tree or "
                      "using --mink and --maxk, the --jaccard flag will be ignored.")
                jaccard = False
        kij_rows, j_rows = [], []
        # every 2-way union of every pair at every k in one GPU launch
        sched = None
        if mink and maxk and len(leaves) > 1:
            sched = self.prefetch_union_cards([[a, b] for i, a in enumerate(leaves) for b in leaves[i + 1:]],
                                              int(mink), min(int(maxk), 64), pair_exp,
                                              schedule=int(maxk) <= 64 and not pair_exp.get("safety"))
        if isinstance(sched, dict):
            # ... and every pair's summary from that table (_FlatUnion): the steps SubSpider + find_delta + kij_summarize +
            # jaccard_summarize take (lib/huffman_dandd.py:685-690, 772-815), in the same order -- the climbs move
            # speciesinfo.kstart, which the next pair starts from --, without a SubSpider, a node and a Sketch per (pair, k)
            table, index, lo, hi = sched["table"], sched["index"], sched["lo"], sched["hi"]
            cards = self.speciesinfo.cardkey
            if jaccard:
                for leaf in leaves:
                    leaf.node_ksweep(mink=mink, maxk=maxk)     # (once per leaf: what every pair's ksweep asks of its two leaves)
                j_rows = RowTable(("A", "B", "Atitle", "Btitle", "kval", "Acard", "Bcard", "ABcard", "jaccard"))
                jc = j_rows.cols
                ks = list(range(mink, maxk + 1))
                leaf_cards = {id(leaf): [leaf.ksketches[k].card for k in ks] for leaf in leaves}
            for i, a in enumerate(leaves):
                for b in leaves[i + 1:]:
                    pair = _FlatUnion(self, [a, b], pair_exp, table[index[id(a)], index[id(b)]], lo, hi)
                    for kk, k in enumerate(range(lo, hi + 1)):   # what the union sketches' cardinalities would have been cached as
                        cards[pair.tmpl.with_k(k)] = float(pair.row[kk])
                    pair.find_delta(self.speciesinfo.kstart)       # SubSpider.__init__ ...
                    pair.fill()
                    pair.find_delta(self.root_k())                 # ... and the climb from the tree's own argmax-k
                    if pair.bestk > 0:
                        pair.touch(pair.bestk)
                    a.update_node(a.bestk)
                    b.update_node(b.bestk)
                    x, y = (a, b) if [a.node_title, b.node_title] == sorted([a.node_title, b.node_title]) else (b, a)
                    row = {"A": x.fastas[0], "B": y.fastas[0], "Adelta": x.delta, "Bdelta": y.delta, "Ak": x.bestk,
                           "Bk": y.bestk, "ABdelta": pair.delta, "ABk": pair.bestk, "Atitle": x.node_title,
                           "Btitle": y.node_title}
                    row["KIJ"] = (row["Adelta"] + row["Bdelta"] - row["ABdelta"]) / row["ABdelta"]
                    kij_rows.append(row)
                    if jaccard:
                        pair.node_ksweep(mink, maxk)
                        # tree order, never swapped (lib/huffman_dandd.py:804); a column at a time
                        ac, bc, abc = leaf_cards[id(a)], leaf_cards[id(b)], [pair.card(k) for k in ks]
                        nk = len(ks)
                        jc["A"] += [a.fastas[0]] * nk
                        jc["B"] += [b.fastas[0]] * nk
                        jc["Atitle"] += [a.node_title] * nk
                        jc["Btitle"] += [b.node_title] * nk
                        jc["kval"] += ks
                        jc["Acard"] += ac
                        jc["Bcard"] += bc
                        jc["ABcard"] += abc
                        jc["jaccard"] += [(x + y - z) / z for x, y, z in zip(ac, bc, abc)]
            return kij_rows, j_rows
        for i, a in enumerate(leaves):
            for b in leaves[i + 1:]:
                pair = SubSpider([a, b], self.speciesinfo, pair_exp)
                pair.root.find_delta(self.root_k())
                kij_rows.append(pair.kij_summarize())
                if jaccard:
                    pair.ksweep(mink=mink, maxk=maxk)
                    j_rows.extend(pair.jaccard_summarize(mink=mink, maxk=maxk))
        return kij_rows, j_rows

    def prepare_AFproject(self, kijsummary, jsummary):
        tool = self.experiment["tool"]
        out = set()
        for d in kijsummary:
            out.add((tool, d["Atitle"], d["Btitle"], 0, d["KIJ"], d["Ak"], d["Bk"], d["ABk"]))
        for d in jsummary:
            out.add((tool, d["Atitle"], d["Btitle"], d["kval"], d["jaccard"], None, None, None))
        return list(out)


def write_phylip(tuples, path, k=0):
    """Lower-triangular PHYLIP distance matrix (distance = 1 - similarity) of the `--afproject` tuples
    (tool, name1, name2, k, value, k1, k2, k12) at one k -- k = 0 selects the K-independent Jaccard rows.  The
    format the reference's helper hands to `fneighbor` (helpers/allpairs.py:182-207: count line, then one row per
    name in sorted order holding the distances to the names before it)."""
    recs, names = {}, set()
    for _tool, a, b, kk, value, _k1, _k2, _k12 in tuples:
        if kk != k:
            continue
        names.update((a, b))
        recs[(a, b)] = recs[(b, a)] = 1 - value
    if not recs:
        raise ValueError(f"no pair at k={k}")
    names = sorted(names)
    with open(path, "wt") as f:
        print(len(names), file=f)
        for i, a in enumerate(names):
            print(" ".join(map(str, [a] + [recs[(a, b)] for b in names[:i]])), file=f)
    return names


class SubSpider(DeltaTree):
    """Existing leaf nodes under one new union node."""

    def __init__(self, leafnodes, speciesinfo, experiment):
        self.speciesinfo = speciesinfo
        self.fastahex = speciesinfo.fastahex
        self.experiment = experiment
        self.kstart = speciesinfo.kstart
        if experiment["ksweep"] is not None:
            self.mink, self.maxk = experiment["ksweep"]
        self._build_tree(leafnodes)
        self.root = self._dt[-1]
        self.fastas = self.root.fastas
        self.ngen = len(self.fastas)
        self.delta = None
        self.fill_tree()
        if experiment["ksweep"] is None:
            self.delta = self.root_delta()
        else:
            self.mink, self.maxk = experiment["ksweep"]

    def _build_tree(self, leafnodes):
        kids = list(leafnodes)
        body = DeltaTreeNode("_".join(os.path.basename(c.node_title) for c in kids), kids, self.speciesinfo,
                             self.experiment, progeny=[leaf for c in kids for leaf in c.progeny])
        if self.experiment["ksweep"] is None:
            body.find_delta(kval=self.speciesinfo.kstart)
        else:
            body.node_ksweep(mink=self.mink, maxk=self.maxk)
        self.mink, self.maxk = body.mink, body.maxk
        self._dt = kids + [body]

    def kij_summarize(self):
        if len(self.fastas) != 2:
            raise ValueError("KIJ can only be calculated on spider/trees with 2 children")
        self.root.update_node(self.root.bestk)
        a, b = self._dt[0], self._dt[1]
        a.update_node(a.bestk)
        b.update_node(b.bestk)
        if [a.node_title, b.node_title] != sorted([a.node_title, b.node_title]):
            a, b = b, a
        row = {"A": a.fastas[0], "B": b.fastas[0], "Adelta": a.delta, "Bdelta": b.delta, "Ak": a.bestk,
               "Bk": b.bestk, "ABdelta": self.root.delta, "ABk": self.root.bestk, "Atitle": a.node_title,
               "Btitle": b.node_title}
        row["KIJ"] = (row["Adelta"] + row["Bdelta"] - row["ABdelta"]) / row["ABdelta"]
        return row

    def jaccard_summarize(self, mink=2, maxk=32):
        if len(self.fastas) != 2:
            raise ValueError("KIJ can only be calculated on spider/trees with 2 or more children")
        a, b = self._dt[0], self._dt[1]  # tree order, never swapped (lib/huffman_dandd.py:804)
        self.ksweep(mink=mink, maxk=maxk)
        rows = []
        for k in range(mink, maxk + 1):
            row = {"A": a.fastas[0], "B": b.fastas[0], "Atitle": a.node_title, "Btitle": b.node_title, "kval": k,
                   "Acard": a.ksketches[k].card, "Bcard": b.ksketches[k].card, "ABcard": self.root.ksketches[k].card}
            row["jaccard"] = (row["Acard"] + row["Bcard"] - row["ABcard"]) / row["ABcard"]
            rows.append(row)
        return rows


class DeltaSpider(DeltaTree):
    """All leaves directly under one root."""

    def __init__(self, fasta_files, speciesinfo, experiment, padding=False):
        super().__init__(fasta_files=fasta_files, speciesinfo=speciesinfo, experiment=experiment,
                         nchildren=len(fasta_files), padding=padding)


def create_delta_tree(tag, genomedir, sketchdir, kstart, nchildren=None, registers=0, flist_loc=None,
                      canonicalize=True, tool="dashing", debug=False, nthreads=0, safety=False, fast=False,
                      verbose=False, ksweep=None, lowmem=False):
    experiment = {"registers": registers, "canonicalize": canonicalize, "tool": tool, "nthreads": int(nthreads),
                  "debug": debug, "baseset": set(), "safety": safety, "fast": fast, "verbose": verbose,
                  "ksweep": ksweep, "lowmem": lowmem}
    speciesinfo = Catalog(tag=tag, genomedir=genomedir, sketchdir=sketchdir, kstart=kstart, tool=tool,
                          flist_loc=flist_loc)
    if flist_loc:
        with open(flist_loc) as f:
            fastas = [line.strip() for line in f]
    elif genomedir and os.path.exists(genomedir):
        fastas = speciesinfo.retrieve_fasta_files(full=True)
    else:
        raise ValueError("You must provide either an existing directory of fastas or a file listing the paths "
                         f"of the desired fastas. The directory you provided was {genomedir}.")
    fastas.sort()
    if nchildren:
        tree = DeltaTree(fasta_files=fastas, speciesinfo=speciesinfo, nchildren=nchildren, experiment=experiment)
    else:
        tree = DeltaSpider(fasta_files=fastas, speciesinfo=speciesinfo, experiment=experiment)
    speciesinfo.save_cardkey(tool=tool, fast=fast)
    speciesinfo.save_references(fast=fast)
    return tree
