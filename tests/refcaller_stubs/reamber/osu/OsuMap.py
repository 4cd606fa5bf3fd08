class OsuMap:
    @staticmethod
    def read_file(path):
        with open(path, encoding="utf8") as f:
            return f.read()
