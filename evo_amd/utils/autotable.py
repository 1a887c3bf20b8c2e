"""AutoTable: append-per-epoch tables on disk (reference: evo/utils/autotable.py).

The reference appends one row per call to an extendable, zlib-compressed HDF5 array through PyTables
(autotable.py:93-131, 232-270): ``tbl.append("F", F)`` grows the array ``/F`` by one row whose shape is the
value's.  This class keeps that interface and, when PyTables is importable, that on-disk format (same node
names, atoms, filters), so a file written here reads back with the reference's own tooling.  PyTables is not
part of this image; without it the rows go to ``<stem>.npz`` (one array per table, rows stacked along axis 0; same
content, NumPy container): a warning names that file once, and the container is rewritten every ``flush_every`` appends
and on ``close()``, so a run that dies keeps all but its last few epochs."""
import os
import sys
import warnings as _warnings

import numpy as np

try:  # the reference's backend
    import tables as _tables
except Exception:  # not installed (this image): NumPy container instead
    _tables = None


class AutoTable:
    def __init__(self, fname=None, compression_level=1, rwmode="w", warnings=True, flush_every=64):
        self.warnings = warnings
        self.flush_every = int(flush_every)
        self._pending = 0
        if fname is None:
            fname = self._guess_fname()
        self.fname = fname
        self.compression_level = compression_level
        self.tables = {}
        self.types = {}
        self._rows = {}  # NumPy container: name -> list of rows
        self.h5 = _tables.open_file(fname, rwmode) if _tables is not None else None
        self.backend = "pytables" if self.h5 is not None else "npz"
        self._closed = False
        if self.h5 is None and warnings:
            _warnings.warn("PyTables is not installed: tables of %r are written to %r (NumPy container)" %
                           (self.fname, self.npz_name()), RuntimeWarning, stacklevel=2)

    def __enter__(self):
        return self

    def __exit__(self, *exc_info):
        self.close()

    @staticmethod
    def _guess_fname():
        base = os.path.splitext(os.path.basename(sys.argv[0] or "evo_amd"))[0]
        return base + ".h5"

    def npz_name(self):
        stem, _ = os.path.splitext(self.fname)
        return stem + ".npz"

    def close(self):
        if self._closed:
            return
        self._closed = True
        if self.h5 is not None:
            self.h5.close()
            return
        self.flush()

    def flush(self):
        """NumPy container: (re)write ``<stem>.npz`` with every row appended so far."""
        if self.h5 is not None:
            self.h5.flush()
            return
        self._pending = 0
        out = {}
        for name, rows in self._rows.items():
            if self.types.get(name) is str:
                out[name] = np.array(rows)
            else:
                out[name] = np.stack(rows, axis=0) if rows else np.zeros((0,))
        np.savez_compressed(self.npz_name(), **out)

    # ---- appending -----------------------------------------------------------------------------
    def _as_array(self, value):
        if isinstance(value, np.ma.MaskedArray):
            value = value.data
        if isinstance(value, str):
            return value
        if np.isscalar(value):
            value = np.asarray(value)
        if not isinstance(value, np.ndarray):
            raise TypeError("Don't know how to handle values of type '%s'" % type(value))
        return value

    def _create_table(self, name, example):
        if self.h5 is not None:
            if isinstance(example, str):
                atom = _tables.VLStringAtom()
            else:
                try:
                    atom = _tables.Atom.from_dtype(example.dtype)
                except Exception:
                    raise TypeError("Could not create table %s because of unknown dtype '%s'" % (name, example.dtype))
            filters = _tables.Filters(complevel=self.compression_level, complib="zlib", shuffle=True)
            if isinstance(example, str):
                self.tables[name] = self.h5.create_vlarray(self.h5.root, name, atom, filters=filters)
            else:
                self.tables[name] = self.h5.create_earray(self.h5.root, name, atom, (0,) + example.shape, filters=filters)
        else:
            self._rows[name] = []
            self.tables[name] = self._rows[name]
        self.types[name] = str if isinstance(example, str) else np.ndarray

    def append(self, name, value):
        """One more row of table ``name`` (autotable.py:93-131)."""
        value = self._as_array(value)
        if name not in self.tables:
            self._create_table(name, value)
        if self.h5 is not None:
            if isinstance(value, str):
                self.tables[name].append(value.encode())
            else:
                try:
                    self.tables[name].append(value.reshape((1,) + value.shape))
                except ValueError:
                    raise TypeError('Wrong datatype "%s" for "%s" field' % (value.dtype, name))
            self.tables[name].flush()
            return
        rows = self._rows[name]
        if not isinstance(value, str) and rows and (rows[0].shape != value.shape):
            raise TypeError('Wrong shape %s for "%s" field (rows are %s)' % (value.shape, name, rows[0].shape))
        rows.append(value if isinstance(value, str) else np.array(value))
        self._pending += 1
        if self.flush_every > 0 and self._pending >= self.flush_every:
            self.flush()

    def assign(self, name, value):
        """Replace table ``name`` by the rows of ``value`` (autotable.py:133-173)."""
        value = self._as_array(value)
        if name in self.tables:
            if self.h5 is not None:
                self.h5.remove_node(self.h5.root, name)
            self.tables.pop(name)
            self._rows.pop(name, None)
        if isinstance(value, str) or value.ndim == 0:
            self.append(name, value)
            return
        for row in value:
            self.append(name, row)

    def append_all(self, valdict):
        for name, value in valdict.items():
            self.append(name, value)
