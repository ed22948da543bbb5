from .configuration import Configuration
from .context import Context, DefaultContext
from .observation import Observation
from .session import Session, OrganicSessions
