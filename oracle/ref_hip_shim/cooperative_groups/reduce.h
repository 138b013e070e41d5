#pragma once
// cooperative_groups::reduce is included by the reference but not used on this path
