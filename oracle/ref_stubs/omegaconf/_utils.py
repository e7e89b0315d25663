def is_primitive_type(t):
    return t in (int, float, bool, str, type(None))
