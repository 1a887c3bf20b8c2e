"""Per-epoch logging: named values are routed to sinks, on rank 0 only.

What training scripts written for the reference call (examples/bars-test/main.py:150-162) keeps working:

    dlog = DataLog(comm)
    dlog.set_handler(("F", "S_nunique"), TextPrinter)     # these two names -> stdout
    dlog.set_handler("*", StoreToH5, "training.h5")       # every name -> one table each, one row per epoch
    dlog.append_all({"F": F, "W": theta["W"]})
    dlog.close()

Behaviour kept from evo/utils/datalog.py:137-274: only rank 0 routes or writes anything; a sink registered for "*"
receives every name; `append_all` hands each sink ONE call with exactly the names it listens to; `ignored(name)` tells a
caller that nobody listens, so an expensive value need not be computed.  The design is this package's own: a routing
table (name -> sinks, plus the wildcard sinks) instead of a policy list that is searched per call, and sinks that only
have to provide `append`.  Off the timed path; StoreToH5 writes through AutoTable (HDF5 when PyTables is installed)."""
import os
import time

from .autotable import AutoTable
from .parallel import SerialComm, pprint

WILDCARD = "*"


class DataHandler:
    """A sink.  Subclasses provide `append(name, value)`; the rest has defaults."""

    def register(self, names):
        """Called once with the name(s) the sink was registered for."""

    def append(self, name, value):
        raise NotImplementedError("%s does not implement append()" % type(self).__name__)

    def append_all(self, values):
        for name in values:
            self.append(name, values[name])

    def assign(self, name, value):
        raise NotImplementedError("%s cannot replace a table" % type(self).__name__)

    def remove(self, name):
        """The sink stops receiving `name`."""

    def close(self):
        """Flush and release whatever the sink holds."""


class StoreToH5(DataHandler):
    """Rows into an AutoTable: a file name opens one, an AutoTable is used as it is, and no argument shares one
    process-wide table between all sinks created that way (what the reference's scripts rely on)."""

    default_autotbl = None

    def __init__(self, destination=None, warnings=True):
        self.destination = destination
        if destination is None:
            if StoreToH5.default_autotbl is None:
                StoreToH5.default_autotbl = AutoTable(warnings=warnings)
            self.autotbl = StoreToH5.default_autotbl
        elif isinstance(destination, AutoTable):
            self.autotbl = destination
        elif isinstance(destination, (str, os.PathLike)):
            self.autotbl = AutoTable(os.fspath(destination), warnings=warnings)
        else:
            raise TypeError("StoreToH5: destination is a file name, an AutoTable or None, not %r" % type(destination).__name__)
        if StoreToH5.default_autotbl is None:
            StoreToH5.default_autotbl = self.autotbl

    def __repr__(self):
        return "StoreToH5(%r)" % (self.autotbl.fname,)

    def append(self, name, value):
        self.autotbl.append(name, value)

    def append_all(self, values):
        self.autotbl.append_all(values)

    def assign(self, name, value):
        self.autotbl.assign(name, value)

    def close(self):
        self.autotbl.close()
        if StoreToH5.default_autotbl is self.autotbl:
            StoreToH5.default_autotbl = None


class StoreToTxt(DataHandler):
    """`name = value` lines into a text file (default: terminal.txt, which must not exist yet)."""

    def __init__(self, destination=None):
        if destination is None:
            destination = "terminal.txt"
            if os.path.exists(destination):
                raise ValueError("StoreToTxt: %s exists already; name another file" % destination)
        self._fh = open(destination, "w")

    def append(self, name, value):
        self._fh.write("%s = %s\n" % (name, value))

    def close(self):
        if not self._fh.closed:
            self._fh.close()


class TextPrinter(DataHandler):
    """`name = value` on rank 0's stdout."""

    def append(self, name, value):
        pprint("\t%s = %s " % (name, value))


class DataLog:
    def __init__(self, comm=None):
        self.comm = SerialComm() if comm is None else comm
        self._routes = {}  # name -> sinks, in registration order
        self._wild = []    # sinks registered for every name
        self._sinks = []   # every sink once, in registration order

    # ---- routing ----------------------------------------------------------------------------
    @property
    def _root(self):
        return self.comm.rank == 0

    def _targets(self, name):
        """Sinks of `name`, registration order, wildcard sinks included."""
        named = self._routes.get(name, ())
        return [s for s in self._sinks if s in named or s in self._wild]

    def ignored(self, name):
        """Nobody listens to `name` (always True away from rank 0, where nothing was registered)."""
        return not self._targets(name)

    def set_handler(self, names, handler_class, *args, **kwargs):
        """Create `handler_class(*args, **kwargs)` and route `names` (a name, an iterable of names, or "*") to it.
        Returns the sink (None away from rank 0)."""
        if not self._root:
            return None
        if not (isinstance(handler_class, type) and issubclass(handler_class, DataHandler)):
            raise TypeError("set_handler: %r is not a DataHandler subclass" % (handler_class,))
        if isinstance(names, str):
            wanted = [names]
        else:
            try:
                wanted = [str(n) for n in names]
            except TypeError:
                raise TypeError("set_handler: names must be a string or an iterable of strings") from None
        sink = handler_class(*args, **kwargs)
        sink.register(names)
        self._sinks.append(sink)
        for n in wanted:
            if n == WILDCARD:
                self._wild.append(sink)
            else:
                self._routes.setdefault(n, []).append(sink)
        return sink

    def remove_handler(self, handler):
        if not self._root:
            return
        if handler not in self._sinks:
            raise ValueError("remove_handler: not a sink of this DataLog")
        self._sinks = [s for s in self._sinks if s is not handler]
        self._wild = [s for s in self._wild if s is not handler]
        for n in list(self._routes):
            self._routes[n] = [s for s in self._routes[n] if s is not handler]
            if not self._routes[n]:
                del self._routes[n]
        handler.close()

    # ---- values -----------------------------------------------------------------------------
    def append(self, name, value):
        if self._root:
            for sink in self._targets(name):
                sink.append(name, value)

    def assign(self, name, value):
        if self._root:
            for sink in self._targets(name):
                sink.assign(name, value)

    def append_all(self, values):
        """One call per sink, with the sub-dict of the names it listens to (insertion order kept)."""
        if not self._root:
            return
        per_sink = {}
        for name, value in values.items():
            for sink in self._targets(name):
                per_sink.setdefault(id(sink), (sink, {}))[1][name] = value
        for sink in self._sinks:
            if id(sink) in per_sink:
                sink.append_all(per_sink[id(sink)][1])

    def progress(self, message, completed=None, width=40):
        """A time-stamped line on rank 0; with `completed` in [0, 1] a bar of `width` cells behind it."""
        if not self._root:
            return
        stamp = time.strftime("%H:%M:%S")
        if completed is None:
            print("[%s] %s" % (stamp, message))
            return
        done = max(0, min(width, int(round(width * float(completed)))))
        print("[%s] %s |%s%s| %3.0f%%" % (stamp, message, "#" * done, "." * (width - done), 100.0 * float(completed)))

    def close(self):
        if self._root:
            for sink in self._sinks:
                sink.close()
            self._sinks, self._wild, self._routes = [], [], {}
