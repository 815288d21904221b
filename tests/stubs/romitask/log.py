import logging


def configure_logger(name, *args, **kwargs):
    return logging.getLogger(name)
