class DataStoreBase:
    def __init__(self, capacity):
        self.capacity = capacity
