// Placeholder until the Torch7 .t7 reader lands.
#include "b2f_host.h"

namespace b2f {

bool load_t7(const std::string &path, std::vector<float> &, bool &, std::string &err)
{
    err = "cannot read '" + path + "': the .t7 reader is not built into this version";
    return false;
}

}  // namespace b2f
