import os


def files_exist(files):
    return len(files) != 0 and all(os.path.exists(f) for f in files)
