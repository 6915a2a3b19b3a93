"""Inert stand-in for `pygame` (not installed here): the reference imports it at module top
level but only touches it inside render()/close(), which the fixture generators never call."""


class _Inert:
    def __getattr__(self, name):
        return _Inert()

    def __call__(self, *a, **k):
        return _Inert()


def __getattr__(name):
    return _Inert()
