def compose(*a, **k):
    raise NotImplementedError


def initialize(*a, **k):
    raise NotImplementedError
