"""Root config (reference: config/config.py:3-7)."""
import os


class Config(object):
    root_dir = os.environ.get("HAMER_ROOT_DIR", os.path.abspath(os.path.dirname(__file__)))


opt = Config()
