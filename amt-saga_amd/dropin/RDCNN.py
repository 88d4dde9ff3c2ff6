"""Drop-in for the reference's RDCNN module: ``from RDCNN import res_net``."""
import _path  # noqa: F401
from amt_saga.rdcnn import res_net  # noqa: F401,E402
