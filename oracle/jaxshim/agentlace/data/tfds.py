def populate_datastore(*a, **k):
    raise NotImplementedError("stand-in")
