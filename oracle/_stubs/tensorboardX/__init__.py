class SummaryWriter:
    def __init__(self, *a, **k):
        pass
    def add_scalar(self, *a, **k):
        pass
