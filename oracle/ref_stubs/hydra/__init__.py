"""Stub of hydra (see omegaconf stub)."""


def main(*a, **k):
    def deco(f):
        return f
    return deco


def compose(*a, **k):
    raise NotImplementedError


def initialize(*a, **k):
    raise NotImplementedError
