"""Stub registry: records registrations, creates nothing."""
registry = {}


def register(id=None, entry_point=None, **kwargs):
    registry[id] = dict(entry_point=entry_point, **kwargs)
