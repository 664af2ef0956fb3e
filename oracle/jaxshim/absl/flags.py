FLAGS = None
