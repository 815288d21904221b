from romitask import FilesetTarget, RomiTask


class ImagesFilesetExists(RomiTask):
    """The 'images' fileset of the scan (romitask.task.ImagesFilesetExists in the real package)."""

    def requires(self):
        return []

    def output(self):
        return FilesetTarget("images")
