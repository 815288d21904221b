"""Test stub of romitask: the RomiTask surface Voxels.run touches over an in-memory
scan shared by every task of a test (``romitask.DB``)."""
import luigi


class _File:
    def __init__(self, fid, array=None, metadata=None):
        self.id = fid
        self.array = array
        self._md = dict(metadata or {})
        self.written = None  # ("npz" | "volume", payload) set by plantdb.io

    def get_metadata(self, key=None, default=None):
        return self._md if key is None else self._md.get(key, default)

    def set_metadata(self, data, value=None):
        if isinstance(data, dict):
            self._md.update(data)
        else:
            self._md[data] = value


class _Fileset:
    def __init__(self, scan, fsid, files=(), metadata=None):
        self.scan = scan
        self.id = fsid
        self._files = list(files)
        self._md = dict(metadata or {})
        self.queries = []

    def get_files(self, query=None):
        self.queries.append(query)
        if not query:
            return list(self._files)
        return [f for f in self._files if all(f.get_metadata(k) == v for k, v in dict(query).items())]

    def get_metadata(self, key=None, default=None):
        return self._md if key is None else self._md.get(key, default)

    def create_file(self, fid):
        f = _File(fid)
        self._files.append(f)
        return f


class _Scan:
    def __init__(self, sid, metadata=None):
        self.id = sid
        self._md = dict(metadata or {})
        self.filesets = {}

    def get_metadata(self, key=None, default=None):
        return self._md if key is None else self._md.get(key, default)

    def fileset(self, fsid):
        if fsid not in self.filesets:
            self.filesets[fsid] = _Fileset(self, fsid)
        return self.filesets[fsid]


class _DB:
    """The scan the tasks of one test work on."""
    scan = None


DB = _DB()


class FilesetTarget:
    def __init__(self, fsid):
        self.fsid = fsid

    def get(self):
        return DB.scan.fileset(self.fsid)


class RomiTask(luigi.Task):
    upstream_task = None
    scan_id = luigi.Parameter(default="")

    def requires(self):
        return self.upstream_task()

    def output(self):
        return FilesetTarget(self.get_task_family())

    def input(self):
        req = self.requires()
        if isinstance(req, dict):
            return {k: t.output() for k, t in req.items()}
        return req.output()

    def output_file(self, file_id=None):
        fid = file_id if file_id is not None else self.get_task_family()
        return self.output().get().create_file(fid)
