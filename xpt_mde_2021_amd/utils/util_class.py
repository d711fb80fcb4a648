"""Small host helpers kept from the reference's utils/util_class.py (names only; :11-13, :52-62)."""
from timeit import default_timer as timer


class WrongInputException(Exception):
    """utils/util_class.py:11-13 -- the reference's error convention for bad options."""


class DurationTime:
    """utils/util_class.py:52-62: `with DurationTime() as t: ...; t.duration` in seconds."""

    def __init__(self):
        self.start = 0.0
        self.duration = 0.0

    def __enter__(self):
        self.start = timer()
        return self

    def __exit__(self, exc_type, exc_value, trace_back):
        self.duration = timer() - self.start
