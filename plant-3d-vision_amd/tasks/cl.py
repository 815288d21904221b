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
    """tasks/cl.py:127-147: ``n = int((max - min) / voxel_size) + 1`` per axis, origin = mins,
    both shifted by the scan's ``displacement`` metadata when present."""
    x_min, x_max = bounding_box["x"]
    y_min, y_max = bounding_box["y"]
    z_min, z_max = bounding_box["z"]
    if displacement is not None:
        x_min += displacement["dx"]
        x_max += displacement["dx"]
        y_min += displacement["dy"]
        y_max += displacement["dy"]
        z_min += displacement["dz"]
        z_max += displacement["dz"]
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


def _exp_clip(vol):
    """``vol = np.exp(vol); vol[vol > 1] = 1.0`` (tasks/cl.py:172-174) with ``np.minimum`` in place of
    the boolean-mask assignment: same values (NaN stays NaN, inf becomes 1), a third of the time on a
    512^3 volume -- the mask and the fancy store cost more than the exponential."""
    out = np.exp(np.asarray(vol))
    if out.ndim == 0:
        return out if not out > 1 else out.dtype.type(1.0)
    np.minimum(out, out.dtype.type(1.0), out=out)
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
    if backprojection_cls is None:
        from ..cl import Backprojection as backprojection_cls
    logger.info(f"Processing a list of {len(masks_files)} mask files...")
    if bounding_box is None:
        logger.critical("Could not obtain valid bounding-box!")
        sys.exit("Error with bounding-box definition!")
    logger.info(f"Bounding-box to use: {bounding_box}")
    if displacement is None:
        logger.warning("No 'displacement' found in scan metadata!")
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
    vol = sc.process_fileset(masks_files, str(camera_metadata), bool(invert))
    if _single_valued(vol):  # tasks/cl.py:168, `len(np.unique(vol)) == 1`
        logger.warning("There is something WRONG with the volume!")

    if log and type == "averaging":  # tasks/cl.py:172-174
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


try:  # the luigi task proper, only where the reference's runtime exists
    import luigi  # type: ignore
    from romitask import RomiTask  # type: ignore
except ImportError:
    Voxels = None
else:
    class Voxels(RomiTask):  # pragma: no cover - needs romitask/plantdb, absent here
        """``plant3dvision.tasks.cl.Voxels`` with the MI355X back-projection (same
        parameters and defaults, tasks/cl.py:79-91)."""
        upstream_task = None
        upstream_mask = luigi.TaskParameter()
        upstream_colmap = luigi.TaskParameter()
        query = luigi.DictParameter(default={})
        camera_metadata = luigi.Parameter(default="colmap_camera")
        voxel_size = luigi.FloatParameter(default=1.0)
        type = luigi.Parameter(default="carving")
        log = luigi.BoolParameter(default=True)
        invert = luigi.BoolParameter(default=False)
        labels = luigi.ListParameter(default=[])
        bounding_box = luigi.DictParameter(default=None)

        def requires(self):
            if self.upstream_colmap.get_task_family() == "Colmap":
                return {"masks": self.upstream_mask(), "colmap": self.upstream_colmap()}
            return {"masks": self.upstream_mask()}

        def run(self):
            from plantdb import io  # type: ignore
            masks_fileset = self.input()["masks"].get()
            masks_files = masks_fileset.get_files(query=self.query)
            bbox = self.bounding_box
            if bbox is None:
                bbox = self.output().get().scan.get_metadata("bounding_box")
            if bbox is None and self.upstream_colmap.get_task_family() == "Colmap":
                bbox = self.input()["colmap"].get().get_metadata("bounding_box")
            try:
                displacement = masks_fileset.scan.get_metadata("displacement")
            except Exception:
                displacement = None
            volume, labels, md = voxels_run(
                masks_files, bbox, voxel_size=self.voxel_size, type=self.type, log=self.log,
                invert=self.invert, labels=self.labels, camera_metadata=self.camera_metadata,
                displacement=displacement,
                fileset_label_names=masks_fileset.get_metadata("label_names", default=None))
            outfile = self.output_file()
            if labels is not None:
                io.write_npz(outfile, volume)
            else:
                io.write_volume(outfile, volume)
            outfile.set_metadata(md)
