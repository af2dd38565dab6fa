"""Pickle interchange with the reference: `dandd progressive|kij` start by unpickling the tree that
`dandd tree` saved (/root/reference/lib/dandd_cmd.py:66,108), and the GLOBALs inside such a pickle are
the reference's module and class names -- huffman_dandd.{DeltaTree,DeltaSpider,SubSpider,DeltaTreeNode},
sketch_classes.{SketchFilePath,DashSketchObj,KMCSketchObj}, species_specifics.SpeciesSpecifics (SURVEY.md
section 5).  The host layer keeps the reference's attribute names, so the objects are interchangeable once
the names resolve:

  load_tree(path)      unpickles a tree written by EITHER program into this package's classes
  dump_tree(obj, f)    writes this package's objects under the REFERENCE's global names, so the reference's
                       own `dandd progressive|kij` (or an older DandD analysis script) can load the file

The names are swapped in only while dump_tree runs (classes and sys.modules are restored afterwards): the
package never squats on `import huffman_dandd` in a process that also has the reference on its path.
"""
import contextlib
import pickle
import sys
import types


def _table():
    from . import deltatree as dt
    from . import store as st
    return {
        ("huffman_dandd", "DeltaTree"): dt.DeltaTree,
        ("huffman_dandd", "DeltaSpider"): dt.DeltaSpider,
        ("huffman_dandd", "SubSpider"): dt.SubSpider,
        ("huffman_dandd", "DeltaTreeNode"): dt.DeltaTreeNode,
        ("sketch_classes", "SketchFilePath"): st.SketchPath,
        ("sketch_classes", "DashSketchObj"): dt.DashSketchObj,
        ("sketch_classes", "KMCSketchObj"): dt.KMCSketchObj,
        ("species_specifics", "SpeciesSpecifics"): st.Catalog,
    }


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        cls = _table().get((module, name))
        if cls is not None:
            return cls
        if (module, name) == ("sketch_classes", "SketchObj"):
            from .deltatree import Sketch
            return Sketch
        return super().find_class(module, name)


def load_tree(path):
    with open(path, "rb") as f:
        return _Unpickler(f).load()


@contextlib.contextmanager
def reference_names():
    """While active, this package's classes pickle under the reference's module/class names."""
    table = _table()
    saved_cls = [(cls, cls.__module__, cls.__qualname__, cls.__name__) for cls in table.values()]
    mods = {}
    for (mod, name), cls in table.items():
        mods.setdefault(mod, types.ModuleType(mod))
        setattr(mods[mod], name, cls)
    saved_mods = {m: sys.modules.get(m) for m in mods}
    try:
        for (mod, name), cls in table.items():
            cls.__module__, cls.__qualname__, cls.__name__ = mod, name, name
        sys.modules.update(mods)
        yield
    finally:
        for cls, mod, qual, name in saved_cls:
            cls.__module__, cls.__qualname__, cls.__name__ = mod, qual, name
        for m, old in saved_mods.items():
            if old is None:
                sys.modules.pop(m, None)
            else:
                sys.modules[m] = old


def dump_tree(obj, f, protocol=pickle.DEFAULT_PROTOCOL):
    with reference_names():
        pickle.dump(obj, f, protocol=protocol)
