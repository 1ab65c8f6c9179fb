def zeros(t):
    if t is not None:
        t.data.fill_(0)
