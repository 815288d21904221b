from romitask import RomiTask


class Colmap(RomiTask):
    """Stand-in for the reference's Colmap task: only its family name and output fileset matter."""

    def requires(self):
        return []
