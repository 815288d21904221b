from romitask import RomiTask


class Masks(RomiTask):
    def requires(self):
        return []
