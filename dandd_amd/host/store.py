"""Sketch naming, on-disk layout and the pickled caches -- the parts of DandD's L2 layer that the
sketching engine must keep byte-compatible so an existing sketch directory keeps working:

  * directory layout  <sketchdir>/ngen<N>/k<K>/<base>.hll          (lib/sketch_classes.py:47-52)
  * leaf base name    <fasta basename>.w.<K>.spacing.<R>            (lib/sketch_classes.py:98-104)
  * union base name   hex(sum of blake2b(fasta_i))[:15]_<R>n<N>k<K>[nc]   (:69-78, :105-110)
  * caches            dandd_fastahex.pickle, dandd_sketchinfo.pickle,
                      <tag>_<tool>_cardinalities.pickle (+ .bkp)     (lib/species_specifics.py:40-89)

(all paths relative to /root/reference).  Only the behaviour is reproduced; the code is this
repo's own.
"""
import hashlib
import os
import pickle
import shutil


def file_digest(path, chunk=1 << 20):
    """blake2b hex digest of a file (same digest as lib/sketch_classes.py:12-18)."""
    h = hashlib.blake2b()
    with open(path, "rb") as f:
        while True:
            b = f.read(chunk)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


class Catalog:
    """Per-sketch-directory persistent state (the reference's SpeciesSpecifics)."""

    def __init__(self, tag, genomedir, sketchdir, kstart, tool, flist_loc=None):
        self.tag = tag
        self.sketchdir = sketchdir
        self.inputdir = genomedir
        self.kstart = kstart
        self.flist_loc = flist_loc
        self.orderings = None
        self.card0 = []
        self.fastahex = {}
        self.sketchinfo = {}
        self.cardkey = {}
        self.update(tool)

    # ---- pickles -----------------------------------------------------------------------------
    @staticmethod
    def _load(path):
        for cand in (path, path + ".bkp"):
            if os.path.exists(cand):
                try:
                    with open(cand, "rb") as f:
                        return pickle.load(f)
                except (pickle.UnpicklingError, EOFError):
                    continue
        return {}

    @staticmethod
    def _store(path, obj):
        with open(path + ".bkp", "wb") as f:
            pickle.dump(obj, f)
        shutil.copyfile(path + ".bkp", path)

    def _p_hex(self):
        return os.path.join(self.sketchdir, "dandd_fastahex.pickle")

    def _p_info(self):
        return os.path.join(self.sketchdir, "dandd_sketchinfo.pickle")

    def _p_card(self, tool):
        return os.path.join(self.sketchdir, f"{self.tag}_{tool}_cardinalities.pickle")

    def update(self, tool):
        self.fastahex = self._load(self._p_hex())
        self.cardkey = self._load(self._p_card(tool))
        self.sketchinfo = self._load(self._p_info())

    def save_references(self, fast=False):
        if not fast:
            self._store(self._p_hex(), self.fastahex)
            self._store(self._p_info(), self.sketchinfo)

    def save_cardkey(self, tool, fast=False):
        if not fast:
            self._store(self._p_card(tool), self.cardkey)

    def retrieve_fasta_files(self, full=True):
        # the reference takes every directory entry (its extension regex is never applied,
        # lib/species_specifics.py:91-97); so does this
        names = list(os.listdir(self.inputdir))
        return [os.path.join(self.inputdir, n) for n in names] if full else names


class SketchPath:
    """Where the sketch of a set of FASTAs at one k lives, and under which name."""

    def __init__(self, filenames, kval, catalog, experiment):
        self.ffiles = list(filenames)
        self.files = sorted(os.path.basename(f) for f in self.ffiles)
        self.ngen = len(self.ffiles)
        ktxt = "{}" if kval == 0 else str(kval)
        self.dir = os.path.join(catalog.sketchdir, f"ngen{self.ngen}", "k" + ktxt)
        self.base = self._name(catalog, kval, ktxt, experiment)
        ext = ".hll" if experiment["tool"] != "kmc" else ""
        self.relative = os.path.join(f"ngen{self.ngen}", f"k{kval}", self.base) + ext
        self.full = os.path.join(self.dir, self.base) + ext
        if kval != 0:
            ensure_dir(self.dir)

    def __repr__(self):
        return f"SketchPath[{self.base}, ngen={self.ngen}, {self.full}]"

    def with_k(self, k):
        """Concrete path of the k-placeholder ('{}') form."""
        return self.full.replace("{}", str(k))

    def at_k(self, k, catalog, experiment):
        """The SketchPath the constructor would build for the same files at k, derived from this
        k-placeholder form (kval 0) by substitution: same names, same catalog registration, without
        re-deriving the file-set digest name (a `kij` over 64 genomes asks for 150 000 of these)."""
        if experiment.get("safety"):
            return SketchPath(self.ffiles, k, catalog, experiment)
        ktxt = str(k)
        new = object.__new__(SketchPath)
        new.ffiles, new.files, new.ngen = self.ffiles, self.files, self.ngen
        new.dir = self.dir.replace("{}", ktxt)
        new.base = self.base.replace("{}", ktxt)
        ext = ".hll" if experiment["tool"] != "kmc" else ""
        new.relative = os.path.join(f"ngen{self.ngen}", f"k{k}", new.base) + ext
        new.full = self.full.replace("{}", ktxt)
        ensure_dir(new.dir)
        if new.base not in catalog.sketchinfo:
            catalog.sketchinfo[new.base] = {"sketchbase": new.base, "files": self.files, "ngen": self.ngen,
                                            "kval": k, "registers": experiment["registers"]}
        return new

    def _digest_sum(self, catalog):
        if self.ngen == 1:
            return file_digest(self.ffiles[0])
        return hex(sum(int(catalog.fastahex[b], 16) for b in self.files))

    def _name(self, catalog, kval, ktxt, experiment):
        key = "".join(self.files)
        if key not in catalog.fastahex:
            catalog.fastahex[key] = self._digest_sum(catalog)
        elif experiment.get("safety"):
            again = self._digest_sum(catalog)
            if again != catalog.fastahex[key]:
                raise RuntimeError(f"Checksum does not match stored value for {key}: {again}, {catalog.fastahex[key]}")
        stored = catalog.fastahex[key]
        registers, canon = experiment["registers"], experiment["canonicalize"]
        if self.ngen == 1:
            if experiment["tool"] != "kmc":
                base = f"{self.files[0]}.w.{ktxt}.spacing.{registers}"
            else:
                base = f"{self.files[0]}_k{ktxt}" + ("" if canon else "nc")
        else:
            base = f"{stored[:15]}_{registers}n{self.ngen}k{ktxt}" + ("" if canon else "nc")
        info = {"sketchbase": base, "files": self.files, "ngen": self.ngen, "kval": kval, "registers": registers}
        if base not in catalog.sketchinfo:
            catalog.sketchinfo[base] = info
        elif experiment.get("safety"):
            old = catalog.sketchinfo[base]
            for k, v in old.items():
                if info[k] != v:
                    raise RuntimeError(f"Duplicate keys but not duplicate values: {base}: (1) {old}, (2) {info}")
        return base


_made_dirs = set()
_seen_sketches = set()


def new_command():
    """What this process remembers about the file system is true for ONE command: a resident `dandd serve` forgets it
    between commands (directories and sketch files may have been removed meanwhile)."""
    _made_dirs.clear()
    _seen_sketches.clear()


def ensure_dir(path):
    """os.makedirs(path, exist_ok=True), once per path and process: the tree code asks for the same
    ngen*/k* directories for every node and every k (a million times in a 32-genome `progressive`)."""
    if path not in _made_dirs:
        os.makedirs(path, exist_ok=True)
        _made_dirs.add(path)


def forget_sketch(path):
    """A sketch file was removed on purpose (Sketch.remove_sketch): drop the remembered answer."""
    _seen_sketches.discard(path)


def sketch_exists(path):
    """A non-empty sketch file is there.  Sketch files are only ever created, never removed or
    truncated, within a run, so a positive answer is remembered; a negative one is asked again."""
    if path in _seen_sketches:
        return True
    try:
        ok = os.stat(path).st_size != 0
    except OSError:
        return False
    if ok:
        _seen_sketches.add(path)
    return ok
