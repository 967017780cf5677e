"""anakin/utils/logger.py: one process-wide logger with info / warning / error."""
import logging

logger = logging.getLogger("anakin")
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("%(asctime)s %(levelname)s %(message)s"))
    logger.addHandler(_h)
    logger.setLevel(logging.INFO)
logger.filehandler = None
