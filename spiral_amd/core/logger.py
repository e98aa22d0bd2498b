"""Process-wide logger of the package (reference: spiro/core/logger.py:30-80): records below
ERROR go to stdout, ERROR and above to stderr, everything from `level_to_file` up also to a
file.  Calling it again re-installs the handlers (it configures the ROOT logger, as the
reference does)."""
import logging
import sys

FORMAT = "%(asctime)s,%(msecs)d %(levelname)-8s [%(pathname)s:%(lineno)d] %(message)s"
DATE_FORMAT = "%Y-%m-%d:%H:%M:%S"


def get_logger(log_file: str = "spiro.log", level_to_file: str = "INFO") -> logging.Logger:
    root = logging.getLogger()
    for old in list(root.handlers):
        root.removeHandler(old)
        old.close()
    fmt = logging.Formatter(FORMAT, datefmt=DATE_FORMAT)

    err = logging.StreamHandler(sys.stderr)
    err.setLevel(logging.ERROR)
    out = logging.StreamHandler(sys.stdout)
    out.setLevel(logging.DEBUG)
    out.addFilter(lambda record: record.levelno < logging.ERROR)
    to_file = logging.FileHandler(log_file, encoding="utf8")
    to_file.setLevel(level_to_file)
    for handler in (err, out, to_file):
        handler.setFormatter(fmt)
        root.addHandler(handler)
    root.setLevel(logging.NOTSET)  # the handlers' own levels decide
    return root
