class GlobalHydra:
    @classmethod
    def instance(cls):
        return cls()

    def is_initialized(self):
        return False

    def clear(self):
        return None
