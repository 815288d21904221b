"""Test stub of plantdb.io: the three calls on the path (read_image, write_npz, write_volume) keep
their payload on the file object instead of a database."""


def read_image(fi):
    return fi.array


def write_npz(fi, data):
    fi.written = ("npz", dict(data))


def write_volume(fi, data):
    fi.written = ("volume", data)
