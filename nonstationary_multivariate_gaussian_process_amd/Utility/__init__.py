"""Drop-in mirror of the reference's ``Utility`` package for the log-posterior path.

Same module names, function names, positional order, keyword names and defaults as
``/root/reference/Utility/{settings,kernels,kronecker_operation,distributions,utils,logpos,prediction}.py``;
the arithmetic runs on the MI355X through libnmgp_hip.so (C ABI in include/nmgp.h).  Put this package's parent
directory first on ``sys.path`` and the reference's model scripts' ``from Utility import logpos`` resolves here
(see INTEGRATION.md).
"""
from . import settings  # noqa: F401
from . import utils  # noqa: F401
from . import kernels  # noqa: F401
from . import kronecker_operation  # noqa: F401
from . import distributions  # noqa: F401
from . import logpos  # noqa: F401
from . import prediction  # noqa: F401
