class ListConfig(list):
    pass
