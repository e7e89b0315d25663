class bitarray(list):
    pass
