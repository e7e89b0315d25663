def read_mat(*a, **k):
    raise NotImplementedError


def write_mat(*a, **k):
    raise NotImplementedError
