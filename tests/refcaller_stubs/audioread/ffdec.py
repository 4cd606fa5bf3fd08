def available():
    return False
