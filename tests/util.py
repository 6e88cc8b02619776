import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REALSHORT = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/realshort.mp4"
COCKATOO = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/cockatoo.mp4"


def have_file(path):
    return os.path.exists(path)
