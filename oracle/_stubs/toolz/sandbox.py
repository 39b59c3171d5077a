def concat(x):
    import itertools
    return itertools.chain.from_iterable(x)
