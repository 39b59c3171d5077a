def set_trace(*a, **k):
    pass
