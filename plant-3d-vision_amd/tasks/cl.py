"""The ``Voxels`` task logic of the reference (``plant3dvision/tasks/cl.py:18-186``) around
the MI355X ``Backprojection``.

``romitask`` / ``plantdb`` / ``luigi`` are unvendored submodules of the reference (empty in
its tree), so the task's *logic* -- bounding box -> shape/origin, label discovery, the
back-projection call, the exp/clip post-processing and the output metadata -- is a plain
function, ``voxels_run``; ``Voxels`` wraps it as a luigi task with the reference's parameter
names and defaults when ``romitask`` is importable.
"""
import logging
import os
import sys

import numpy as np

logger = logging.getLogger(__name__)

#: parameter defaults of the reference task (tasks/cl.py:83-91)
VOXELS_DEFAULTS = dict(query={}, camera_metadata="colmap_camera", voxel_size=1.0, type="carving",
                       log=True, invert=False, labels=[], bounding_box=None)


def grid_from_bounding_box(bounding_box, voxel_size, displacement=None):
    """tasks/cl.py:125-147: ``n = int((max - min) / voxel_size) + 1`` per axis, origin = mins, both
    shifted by the scan's ``displacement`` metadata.  The shifts are applied one by one inside a
    ``try`` like the reference's (:129-140): no metadata (``None``) shifts nothing, a dictionary
    lacking a key shifts the bounds before it and warns."""
    x_min, x_max = bounding_box["x"]
    y_min, y_max = bounding_box["y"]
    z_min, z_max = bounding_box["z"]
    try:
        x_min += displacement["dx"]
        x_max += displacement["dx"]
        y_min += displacement["dy"]
        y_max += displacement["dy"]
        z_min += displacement["dz"]
        z_max += displacement["dz"]
    except Exception:  # the reference's bare except (:139)
        logger.warning("No 'displacement' found in scan metadata!")
    nx = int((x_max - x_min) / voxel_size) + 1
    ny = int((y_max - y_min) / voxel_size) + 1
    nz = int((z_max - z_min) / voxel_size) + 1
    return [nx, ny, nz], [x_min, y_min, z_min]


def _single_valued(vol):
    """``len(np.unique(vol)) == 1`` (tasks/cl.py:168) without sorting the volume: np.unique on a
    512^3 grid costs seconds, this a look at the first few values in the usual case and one pass
    otherwise.  NaNs count as one value, as in np.unique (``equal_nan=True``)."""
    flat = np.asarray(vol).reshape(-1)
    if flat.size == 0:
        return False
    first = flat[0]
    if first != first:  # NaN
        return bool(np.isnan(flat[:4096]).all() and np.isnan(flat).all())
    return bool((flat[:4096] == first).all() and (flat == first).all())


def _exp_clip(vol, workers=None, inplace=False):
    """``vol = np.exp(vol); vol[vol > 1] = 1.0`` (tasks/cl.py:172-174) with ``np.minimum`` in place of
    the boolean-mask assignment: same values (NaN stays NaN, inf becomes 1), a third of the time on a
    512^3 volume -- the mask and the fancy store cost more than the exponential.  Volumes of 2^22
    elements and more are done slab by slab on a few host threads (elementwise, so the same bits).
    ``inplace`` (for a caller that owns ``vol``) writes the result over the input: no second
    volume-sized array to fault in, none to hand back to the OS (25 ms for 512 MiB on the GPU box)."""
    src = np.asarray(vol)
    if src.ndim == 0:
        out = np.exp(src)
        return out if not out > 1 else out.dtype.type(1.0)
    inplace = bool(inplace) and src.dtype.kind == "f" and src.flags.writeable
    if src.size < (1 << 22) or not src.flags.c_contiguous or src.dtype.kind != "f":
        out = np.exp(src, out=src) if inplace else np.exp(src)
        np.minimum(out, out.dtype.type(1.0), out=out)
        return out
    import threading
    from .. import _native as nat
    out = src if inplace else np.empty_like(src)
    fs, fo = src.reshape(-1), out.reshape(-1)
    parts = max(1, min(int(workers or nat.host_workers()), src.size >> 20))
    bounds = [src.size * q // parts for q in range(parts + 1)]
    one = out.dtype.type(1.0)

    def work(a, b):
        np.exp(fs[a:b], out=fo[a:b])  # NumPy releases the interpreter lock in both loops
        np.minimum(fo[a:b], one, out=fo[a:b])

    threads = [threading.Thread(target=work, args=(bounds[q], bounds[q + 1])) for q in range(parts)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    return out


def voxels_run(masks_files, bounding_box, voxel_size=1.0, type="carving", log=True, invert=False,
               labels=(), camera_metadata="colmap_camera", displacement=None,
               fileset_label_names=None, device=0, backprojection_cls=None):
    """``Voxels.run`` without luigi/plantdb (tasks/cl.py:99-186).

    masks_files : list of file-like objects (``.id``, ``.get_metadata(key, default=None)``,
        pixels via ``plantdb.io.read_image`` or ``.read_image()`` / ``.array``).
    bounding_box : ``{'x': [min, max], 'y': ..., 'z': ...}``; ``None`` is the reference's
        hard failure (``sys.exit``, tasks/cl.py:120-122).
    fileset_label_names : the masks fileset's ``label_names`` metadata, used when ``labels``
        is empty (tasks/cl.py:149-156).

    Returns ``(volume, labels, metadata)``: ``volume`` is what the reference writes -- a dict
    ``{label: array}`` (NPZ, tasks/cl.py:176-182) when labels are in play, else the single
    array (``write_volume``, :184) -- and ``metadata = {'voxel_size', 'origin'}`` (:186).
    """
    if backprojection_cls is None:  # the product: the HIP engine (an argument only: the CPU tests pass a class of theirs)
        from ..cl import Backprojection as backprojection_cls
    logger.info(f"Processing a list of {len(masks_files)} mask files...")
    if bounding_box is None:
        logger.critical("Could not obtain valid bounding-box!")
        sys.exit("Error with bounding-box definition!")
    logger.info(f"Bounding-box to use: {bounding_box}")
    shape, origin_list = grid_from_bounding_box(bounding_box, voxel_size, displacement)
    origin = np.array(origin_list)

    if len(labels) == 0:
        use_labels = fileset_label_names
        if use_labels is None or len(use_labels) == 0:
            logger.warning("No metadata 'label_names' in `masks_fileset`!")
    else:
        use_labels = list(labels)

    sc = backprojection_cls(shape=shape, origin=origin_list, voxel_size=float(voxel_size),
                            type=str(type), labels=use_labels, log=bool(log), device=device)
    want_exp = bool(log) and type == "averaging"
    if want_exp and use_labels is not None and hasattr(sc, "_label_post"):
        # our own class: the labelled read-back applies np.exp + clip piece by piece on its way into the float64 array
        # (Backprojection._process_labels_staged) -- announced here, confirmed by `_label_post_applied`
        sc._label_post = "exp_clip"
    vol = sc.process_fileset(masks_files, str(camera_metadata), bool(invert))
    staged = bool(getattr(sc, "_label_post_applied", False))
    single = getattr(sc, "_label_single_valued", None)  # (the staged read-back has looked, before its exponential)
    if single is None:
        single = _single_valued(vol)
    if single:  # tasks/cl.py:168, `len(np.unique(vol)) == 1` (looked at before the exponential, as there)
        logger.warning("There is something WRONG with the volume!")

    if want_exp and not staged:  # tasks/cl.py:172-174
        vol = _exp_clip(vol)

    if use_labels is not None:
        out = {}
        for i, label in enumerate(use_labels):
            out[label] = vol[i, :]
        volume = out
    else:
        volume = vol
    metadata = {"voxel_size": voxel_size, "origin": origin.tolist()}
    if hasattr(sc, "close"):
        sc.close()
    return volume, use_labels, metadata


try:  # the luigi task proper, where the reference's runtime (or the test stubs) can be imported
    import luigi  # type: ignore
    from romitask import RomiTask  # type: ignore
    from romitask.task import ImagesFilesetExists  # type: ignore
except ImportError:
    Voxels = None
else:
    try:  # the reference's own upstream tasks are the parameter defaults (tasks/cl.py:79-80)
        from plant3dvision.tasks.colmap import Colmap  # type: ignore
        from plant3dvision.tasks.proc2d import Masks  # type: ignore
    except ImportError:
        Colmap = Masks = None

    def _task_parameter(default):
        return luigi.TaskParameter(default=default) if default is not None else luigi.TaskParameter()

    class Voxels(RomiTask):
        """``plant3dvision.tasks.cl.Voxels`` with the MI355X back-projection: same parameters and
        defaults (tasks/cl.py:78-91), same ``requires`` (:93-97), same sources of the bounding box in
        the same order (:106-122), same output (:176-186)."""
        upstream_task = None  # override default attribute from ``RomiTask``   (:78)
        upstream_mask = _task_parameter(Masks)      # :79
        upstream_colmap = _task_parameter(Colmap)   # :80

        query = luigi.DictParameter(default={})                      # :82
        camera_metadata = luigi.Parameter(default='colmap_camera')   # :83
        voxel_size = luigi.FloatParameter(default=1.0)               # :84
        type = luigi.Parameter(default="carving")                    # :85
        log = luigi.BoolParameter(default=True)                      # :86

        invert = luigi.BoolParameter(default=False)                  # :88
        labels = luigi.ListParameter(default=[])                     # :89
        bounding_box = luigi.DictParameter(default=None)             # :90

        def requires(self):
            if self.upstream_colmap.get_task_family() == 'Colmap':
                return {'masks': self.upstream_mask(), 'colmap': self.upstream_colmap()}
            else:
                return {'masks': self.upstream_mask()}

        def run(self):
            from plantdb import io  # type: ignore
            masks_fileset = self.input()['masks'].get()
            masks_files = masks_fileset.get_files(query=self.query)

            # - the bounding box, from (1) the parameter, (2) the scan metadata, (3) the Colmap
            #   fileset, (4) the 'images' fileset (:106-118)
            if self.bounding_box is None:
                self.bounding_box = self.output().get().scan.get_metadata("bounding_box")
                logger.debug(f"Bounding-box from scan metadata: {self.bounding_box}")
            if self.bounding_box is None and self.upstream_colmap.get_task_family() == 'Colmap':
                colmap_fileset = self.input()['colmap'].get()
                if self.bounding_box is None:
                    self.bounding_box = colmap_fileset.get_metadata("bounding_box")
                logger.debug(f"Bounding-box from Colmap fileset: {self.bounding_box}")
            if self.bounding_box is None:
                self.bounding_box = ImagesFilesetExists().output().get().get_metadata("bounding_box")
            if self.bounding_box is None:
                logger.critical(f"Could not obtain valid bounding-box for {self.scan_id}!")

            try:  # :129-140 -- any failure means "no displacement"
                displacement = masks_fileset.scan.get_metadata("displacement")
            except Exception:
                displacement = None
            volume, labels, md = voxels_run(
                masks_files, None if self.bounding_box is None else dict(self.bounding_box),
                voxel_size=self.voxel_size, type=self.type, log=self.log, invert=self.invert,
                labels=self.labels, camera_metadata=self.camera_metadata, displacement=displacement,
                fileset_label_names=masks_fileset.get_metadata("label_names", default=None))

            outfile = self.output_file()
            if labels is not None:
                logger.debug(f"Writing NPZ volume for label: {labels}")
                io.write_npz(outfile, volume)   # :176-182
            else:
                io.write_volume(outfile, volume)  # :184
            outfile.set_metadata(md)  # :186
