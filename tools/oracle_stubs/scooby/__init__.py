"""Stand-in for scooby: an empty Report class."""
__version__ = "0.0.stub"


class Report:
    def __init__(self, *args, **kwargs):
        pass
