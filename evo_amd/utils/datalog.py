"""DataLog: route named per-epoch values to handlers on rank 0 (reference: evo/utils/datalog.py).

    dlog = DataLog(comm)
    dlog.set_handler(("F", "S_nunique"), TextPrinter)
    dlog.set_handler("*", StoreToH5, "training.h5")
    dlog.append_all({"F": F, "W": theta["W"]})      # examples/bars-test/main.py:150-162
    dlog.close()

Same calls, same rank-0-only behaviour (datalog.py:169-175); StoreToH5 writes through AutoTable (HDF5 via PyTables
when that is installed, a NumPy container otherwise -- see autotable.py).  Off the timed path."""
from os.path import isfile
from time import strftime

from .autotable import AutoTable
from .parallel import SerialComm, pprint


class DataHandler:
    """Base class of everything DataLog can hand values to (datalog.py:24-48)."""

    def register(self, tblname):
        pass

    def append(self, tblname, value):
        raise NotImplementedError

    def append_all(self, valdict):
        for key, val in valdict.items():
            self.append(key, val)

    def remove(self, tblname):
        pass

    def close(self):
        pass


class StoreToH5(DataHandler):
    default_autotbl = None

    def __init__(self, destination=None, warnings=True):
        """``destination``: file name, an AutoTable, or None (the process-wide default table, datalog.py:51-77)."""
        self.destination = destination
        if isinstance(destination, AutoTable):
            self.autotbl = destination
        elif isinstance(destination, str):
            self.autotbl = AutoTable(destination, warnings=warnings)
        elif destination is None:
            self.autotbl = StoreToH5.default_autotbl or AutoTable(warnings=warnings)
        else:
            raise TypeError("Expects an AutoTable instance or a string as argument")
        if StoreToH5.default_autotbl is None:
            StoreToH5.default_autotbl = self.autotbl

    def __repr__(self):
        return "StoreToH5 into file %s" % self.destination

    def append(self, tblname, value):
        self.autotbl.append(tblname, value)

    def append_all(self, valdict):
        self.autotbl.append_all(valdict)

    def assign(self, tblname, value):
        self.autotbl.assign(tblname, value)

    def close(self):
        self.autotbl.close()
        if StoreToH5.default_autotbl is self.autotbl:
            StoreToH5.default_autotbl = None


class StoreToTxt(DataHandler):
    def __init__(self, destination=None):
        """``name = value`` lines into a text file (datalog.py:95-122)."""
        if destination is None:
            if isfile("terminal.txt"):
                raise ValueError("Please enter a file name that does not already exist.")
            destination = "terminal.txt"
        self.txt_file = open(destination, "w")

    def append(self, tblname, value):
        self.txt_file.write("%s = %s\n" % (tblname, value))

    def close(self):
        self.txt_file.close()


class TextPrinter(DataHandler):
    def append(self, tblname, value):
        pprint("\t%s = %s " % (tblname, value))

    def append_all(self, valdict):
        for name, val in valdict.items():
            pprint("\t%s = %s \n" % (name, val), end="")


class DataLog:
    def __init__(self, comm=None):
        self.comm = SerialComm() if comm is None else comm
        self.policy = []  # ordered (table name, handler) pairs
        self._lookup_cache = {}

    def _lookup(self, tblname):
        if tblname not in self._lookup_cache:
            self._lookup_cache[tblname] = [h for name, h in self.policy if name == tblname or name == "*"]
        return self._lookup_cache[tblname]

    def progress(self, message, completed=None):
        if self.comm.rank != 0:
            return
        if completed is None:
            print("[%s] %s" % (strftime("%H:%M:%S"), message))
        else:
            totlen = 65 - len(message)
            barlen = int(totlen * completed)
            print("[%s] %s [%s%s]" % (strftime("%H:%M:%S"), message, "*" * barlen, "-" * (totlen - barlen)))

    def append(self, tblname, value):
        if self.comm.rank != 0:
            return
        for h in self._lookup(tblname):
            h.append(tblname, value)

    def assign(self, tblname, value):
        if self.comm.rank != 0:
            return
        for h in self._lookup(tblname):
            h.assign(tblname, value)

    def append_all(self, valdict):
        """Every handler gets the sub-dict of the tables it is registered for (datalog.py:183-207)."""
        if self.comm.rank != 0:
            return
        handlers = []
        for tblname in valdict:
            for h in self._lookup(tblname):
                if h not in handlers:
                    handlers.append(h)
        for h in handlers:
            h.append_all({name: val for name, val in valdict.items() if h in self._lookup(name)})

    def ignored(self, tblname):
        """True when nobody listens to ``tblname``: collecting the value can be skipped (datalog.py:209-226)."""
        return self._lookup(tblname) == []

    def set_handler(self, tblname, handler_class, *args, **kargs):
        if self.comm.rank != 0:
            return None
        if not (isinstance(handler_class, type) and issubclass(handler_class, DataHandler)):
            raise TypeError("handler_class must be a subclass of DataHandler ")
        handler = handler_class(*args, **kargs)
        handler.register(tblname)
        if isinstance(tblname, str):
            self.policy.append((tblname, handler))
        elif hasattr(tblname, "__iter__"):
            for t in tblname:
                self.policy.append((t, handler))
        else:
            raise TypeError("Table-name must be a string (or a list of strings)")
        self._lookup_cache = {}
        return handler

    def remove_handler(self, handler):
        if self.comm.rank != 0:
            return
        if not isinstance(handler, DataHandler):
            raise ValueError("Please provide valid DataHandler object.")
        self.policy = [(n, h) for n, h in self.policy if h is not handler]
        handler.close()
        self._lookup_cache = {}

    def close(self):
        if self.comm.rank != 0:
            return
        closed = []
        for _, handler in self.policy:
            if handler not in closed:
                handler.close()
                closed.append(handler)


