// oracle/ball_ref_driver.cpp -- TEST INFRASTRUCTURE: a command-line front end that compiles the REFERENCE's own
// neighbourhood program (barcode_analysis/5_steps_neibourhoods/{neibourhoods.cpp, AC_UTILS_no_hash.cpp}) from the
// sources where they lie under /root/reference; nothing of the reference is copied here.  Built only in the build
// container (oracle/Makefile target _ref/ball_ref, output under oracle/_ref/); used to pin the C restatement
// (ac_ball_oracle.c) and to produce tests/golden/ball_sizes.json.
//
//   ball_ref <radius> <classic 0|1>  < presentations (one Python-style list per line)  > sizes (one per line)
#define main reference_main  // the reference file has its own main(): it is compiled but not used
#include "/root/reference/barcode_analysis/5_steps_neibourhoods/neibourhoods.cpp"
#undef main

int main(int argc, char** argv) {
    const size_t radius = argc > 1 ? (size_t)atoi(argv[1]) : 5;
    const bool classic = argc > 2 && atoi(argv[2]) != 0;
    std::string line;
    while (std::getline(std::cin, line)) {
        std::vector<int> data;
        std::stringstream ss(line);
        int number;
        char c;
        ss >> c;  // '['
        while (ss >> number) {
            data.push_back(number);
            ss >> c;  // ',' or ']'
        }
        if (data.empty()) continue;
        Relator r1, r2;
        for (size_t i = 0; i < data.size() / 2; i++)
            if (data[i] != 0) r1.push_back(data[i]);
        for (size_t i = data.size() / 2; i < data.size(); i++)
            if (data[i] != 0) r2.push_back(data[i]);
        std::cout << neibourhood(sort_(r1, r2), radius, classic) << std::endl;
    }
    return 0;
}
