class Discrete:
    def __init__(self, n):
        self.n = n
